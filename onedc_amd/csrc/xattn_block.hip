// xattn_block.hip -- the cross-attention half of a BasicTransformerBlock as ONE kernel (round 6).
//
// diffusers BasicTransformerBlock, as the reference's UNet runs it (src/models/sd15_onedc_codec_stage1/decoder_unet.py:204-293):
//     h = h + attn2.to_out( softmax( attn2.to_q(norm2(h)) . K^T / sqrt(d) ) . V )          K, V: projections of the 144 context tokens
// Per pixel everything here is local -- a LayerNorm over its 320 channels, two 320 x 320 projections and an 8-head attention over 144
// keys whose K / V^T (one head of one picture: 11.5 + 11.5 KB) fit LDS -- but as launches it was four full passes over the 94 MB
// residual stream of the UNet's 96 x 96 level (LayerNorm 34 us, to_q 66, attention 113, to_out + residual 97 per block and batch of 16:
// profiles/r5_unet_layers.txt, r5_bench_attn.txt, r5_bench_gn.txt), each bound by HBM and by per-workgroup fixed cost, not by the matrix
// cores.  One workgroup here keeps a tile of 64 pixels resident from the LayerNorm to the residual add:
//   P0  h tile (64 pixels x 640 B) -> registers, LayerNorm (two-pass over the registers, the arithmetic of norm.hip's layernorm_kernel;
//       the host may fold the norm's affine into to_q -- Wq . diag(gamma), bias Wq . beta -- and pass ln_g = NULL: the tile is then only
//       centred and scaled), normalised tile -> LDS image X in the MFMA B-operand layout of the window kernel (64 B per pixel and
//       32-channel chunk, granule g at g ^ 2 * ((px >> 2) & 1)); the first two K-steps of Wq are already on their way (LDS-DMA)
//   P1  q = Wq . X: 10 K-steps of 32 channels; the weights of a K-step (320 rows x 64 B) ride a ring of two LDS slots filled by LDS-DMA,
//       each wave staging exactly the 80 rows its own fragments read -- the ring is wave-private, the K loop has NO barrier --
//       (buffer_load ... lds, granule == lane on both sides: conflict-free fragment reads) and the fragments are double-buffered in
//       registers (those of K-step t + 1 are read while K-step t multiplies, the slot they leave is restaged for t + 3); wave w owns
//       channels 80 w .. 80 w + 79 of all 64 pixels (20 accumulator tiles, 9 fragment reads per 20 MFMAs)
//   P2  q -> X (overwrites the normalised tile in place, same layout: q of head h is granules 5 h .. 5 h + 4 of a pixel)
//   P3  attention, wave w owns pixels 16 w .. 16 w + 15, heads one after the other: K_h / V_h^T of the picture are staged through
//       registers into LDS (the ring is idle) while the previous head multiplies; S^T = K_h . q_h^T (keys on accumulator rows, the
//       pixel on the lane: the softmax is lane-local up to a two-step butterfly), P^T straight from the accumulators as the B operand
//       of O^T += V_h^T . P^T in the permuted-but-consistent key order of attention.hip; o_h -> X in place of q_h
//   P4  y = Wo . X + bias + h (the residual re-read from L2 / HBM in accumulator layout), as P1
//   P5  outputs through LDS, stored as whole 640-byte rows
// Two workgroups (4 waves, 80 KB) per CU: one's LayerNorm / attention / epilogue phases run beside the other's projections.
// HBM traffic per tile: h read twice, y written once -- 283 MB per launch at 96 x 96, B = 16, instead of 850 MB.
// Arithmetic contract: f32 accumulation everywhere, q / o / y rounded to the build's 16-bit format where the unfused path stores them
// (q, o, y), P rounded to bf16 or the build's format exactly as the flash kernels do for the dtype V^T arrives in; the LayerNorm output
// is rounded once (as the unfused path's stored tensor is).  Not bit-identical to the unfused launches (other summation order inside the
// projections' K loop than conv_lin_kernel, plain softmax with the row maximum instead of flash_attn64's lazy one): the test bound is
// the layer tolerance of tests/test_gpu_kernels.py, and the full-size parity tests run through it.
// Shapes: C = 320, 8 heads of 40 (the SD1.5 UNet's 96 x 96 level), Nk a multiple of 16 up to 160, compact heads (layers.COMPACT_HEADS).
#include <type_traits>

#include "conv_common.h"

namespace {

constexpr int XC = 320, XM = 64, XHEADS = 8, XD = 40;
constexpr int X_OP = (XC / 32) * XM * 64;            // 40960: the pixel operand image [chunk][pixel][64 B]
constexpr int X_WT = XC * 64;                        // 20480: weights of one K-step
constexpr int X_RING = X_OP;                         // two slots behind the operand image
constexpr int X_LDS = X_OP + 2 * X_WT;               // 81920: two workgroups per CU
constexpr int X_KROW = 96, X_VROW = 336;             // padded rows of the K_h / V_h^T images (conflict-free fragment reads)
constexpr int X_KIMG = X_RING, X_VIMG = X_RING + 160 * X_KROW;      // 15360 + 40 x 336 = 28800 B over the idle ring
constexpr int X_SROW = 656;                          // epilogue staging: one pixel's 640 B + 16

__device__ __forceinline__ int x_addr(int c8, int px) {         // byte address of the granule of channels c8 .. c8 + 7 (c8 % 8 == 0)
    return (c8 >> 5) * (XM * 64) + px * 64 + ((((c8 >> 3) & 3) ^ (((px >> 2) & 1) << 1)) << 4);
}

#define XA_RD(dst, addr, off)                                                                                 \
    do {                                                                                                      \
        a16x8 &d__ = (dst);                                                                                   \
        const int a__ = (addr);                                                                               \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d__) : "v"(a__), "n"(off));                        \
    } while (0)

#ifdef ONEDC_ACT_F16
#define XA_MFMA_Z(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b))
#else
#define XA_MFMA_Z(d, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b))
#endif
#define XA_MFMA_Z_BF16(d, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b))

// KT: key tiles of 16 known at compile time (9: the 144 context tokens of a 768 x 768 picture), or 0: read from the parameters
template <bool VT_BF16, int KT_>
__global__ __launch_bounds__(256, 2) void xattn_block_kernel(const onedc_xattn_params p, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * XM;                  // first pixel of the tile; a tile lies in one picture (HW % 64 == 0)
    const int img = (int)(row0 / p.HW);
#ifdef EXP_XA_TIMELINE                                               // (diagnostic build: s_memtime at the phase boundaries, written over the
    unsigned long long tl[8];                                        // tile's first output row -- tools/timeline_xattn.py; results are garbage)
#define XA_TL(i) tl[i] = __builtin_amdgcn_s_memtime()
#else
#define XA_TL(i)
#endif
    XA_TL(0);

    // ---- the two projections: acc[i][pt] = channels 80 wave + 16 i .. + 15 (rows) x pixels 16 pt .. + 15 (columns) ------------------
    f32x4 acc[5][4];
    // DMA lane: row (l & 15) of a 16-row block, granule l >> 4.  A wave stages exactly the five row blocks ITS fragments read (rows 80 w ..):
    // the weight ring is private to the wave, its K loop needs no barrier -- only its own counted vmcnt and lgkmcnt
    const unsigned w_voff = (unsigned)(((wave * 80 + r16) * XC + g * 8) * 2);
    const int a_rd = X_RING + wave * 5 * 1024 + lane * 16;                        // fragment lane == DMA lane
    int b_rd[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) b_rd[pt] = (16 * pt + r16) * 64 + ((g ^ ((((16 * pt + r16) >> 2) & 1) << 1)) << 4);
    __amdgpu_buffer_rsrc_t w_rsrc;
    // weights of K-step ks into ring slot `slot`: this wave's row blocks 5 wave .. 5 wave + 4
    auto stage = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < 5; ++jj)
            glds16_buf(w_rsrc, w_voff, (unsigned)(jj * 16 * XC * 2 + ks * 64), smem + X_RING + slot * X_WT + (5 * wave + jj) * 1024);
    };
    auto gemm_issue = [&](const void *wbase) __attribute__((always_inline)) {       // the first two K-steps on their way (the ring must be idle)
        w_rsrc = make_rsrc(wbase, (unsigned)(384 * XC * 2));
        stage(0, 0);
        stage(1, 1);
    };
    a16x8 fa[2][5], fb[2][4];
    auto read_frags = [&](int ks, auto SET) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        const int a0 = a_rd + (ks & 1) * X_WT, b0 = ks * (XM * 64);
#pragma unroll
        for (int i = 0; i < 5; ++i) XA_RD(fa[set][i], a0, i * 1024);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) XA_RD(fb[set][pt], b_rd[pt] + b0, 0);
    };
#define XA_WAIT_FRAGS(set)                                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                              \
                 : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]), "+v"(fa[set][4]), "+v"(fb[set][0]),        \
                   "+v"(fb[set][1]), "+v"(fb[set][2]), "+v"(fb[set][3]))
    // K-step ks on fragment set CUR while the fragments of ks + 1 are read into the other set and their slot is restaged for ks + 3
    auto gemm_step = [&](int ks, auto CUR) __attribute__((always_inline)) {
        constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
        constexpr int NKS = XC / 32;
        if (ks + 1 < NKS) {
            // this wave's pieces of K-step ks + 1 have landed (the five of ks + 2 may fly on): nobody else writes or reads them
            if (ks + 2 < NKS) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            read_frags(ks + 1, std::integral_constant<int, nxt>());
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[i][pt] = ONEDC_MFMA16(fa[cur][i], fb[cur][pt], acc[i][pt]);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < NKS) {
            XA_WAIT_FRAGS(nxt);                               // the fragments of ks + 1 are in registers: their slot may be restaged
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 3 < NKS) stage(ks + 3, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto gemm_run = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[i][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // K-step 0's pieces (the five of K-step 1 may fly on); its fragments; slot 0 restaged for K-step 2
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, std::integral_constant<int, 0>());
        XA_WAIT_FRAGS(0);
        __builtin_amdgcn_sched_barrier(0);
        stage(2, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int ks = 0; ks < XC / 32; ks += 2) {
            gemm_step(ks, std::integral_constant<int, 0>());
            gemm_step(ks + 1, std::integral_constant<int, 1>());
        }
    };
    gemm_issue(p.wq);                                         // (flies while the tile is loaded and normalised)

    // ---- P0: LayerNorm of the tile, four threads per pixel (80 channels each) --------------------------------------------------
    {
        const int px = tid >> 2, part = tid & 3;
        const a16_t *src = (const a16_t *)p.h + (row0 + px) * p.h_cs + p.h_co + part * 80;
        uint4 raw[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) raw[j] = *(const uint4 *)(src + j * 8);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const uint32_t w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) s += a2f_lo(w[e]) + a2f_hi(w[e]);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        const float mean = s / XC;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const uint32_t w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = a2f_lo(w[e]) - mean, d1 = a2f_hi(w[e]) - mean;
                q += d0 * d0;
                q += d1 * d1;
            }
        }
        q += __shfl_xor(q, 1);
        q += __shfl_xor(q, 2);
        const float rstd = rsqrtf(q / XC + p.ln_eps);
        if (p.ln_g) {                                         // the norm's affine here (wave-uniform: a kernel argument)
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int c8 = part * 80 + j * 8;
                const float4 g0 = *(const float4 *)(p.ln_g + c8), g1 = *(const float4 *)(p.ln_g + c8 + 4);
                const float4 b0 = *(const float4 *)(p.ln_b + c8), b1 = *(const float4 *)(p.ln_b + c8 + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                const uint32_t w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = (a2f_lo(w[e]) - mean) * rstd * gg[2 * e] + bb[2 * e];
                    const float v1 = (a2f_hi(w[e]) - mean) * rstd * gg[2 * e + 1] + bb[2 * e + 1];
                    o[e] = pack_a2(v0, v1);
                }
                *(uint4 *)(smem + x_addr(c8, px)) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        } else {                                              // ... or folded into to_q by the host: (x - mean) * rstd only
            const float nm = -mean * rstd;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const uint32_t w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack_a2(__builtin_fmaf(a2f_lo(w[e]), rstd, nm), __builtin_fmaf(a2f_hi(w[e]), rstd, nm));
                *(uint4 *)(smem + x_addr(part * 80 + j * 8, px)) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }

    __syncthreads();                                          // the normalised tile is in LDS
    XA_TL(1);
    gemm_run();
    __syncthreads();                                          // every wave is past its last read of the normalised tile
    XA_TL(2);

    // ---- P2: q -> X ------------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int c = 80 * wave + 16 * i + 4 * g, px = 16 * pt + r16;
            f32x4 a = acc[i][pt];
            if (p.bq) {                                       // (the LayerNorm's beta through to_q, when the host folded the affine)
                const float4 bq = *(const float4 *)(p.bq + c);
                a[0] += bq.x;
                a[1] += bq.y;
                a[2] += bq.z;
                a[3] += bq.w;
            }
            *(uint2 *)(smem + x_addr(c & ~7, px) + ((c >> 2) & 1) * 8) = make_uint2(pack_a2(a[0], a[1]), pack_a2(a[2], a[3]));
        }

    // ---- P3: attention; wave w owns pixels 16 w .. 16 w + 15 ---------------------------------------------------------------------------
    {
        const int KT = KT_ ? KT_ : (p.Nk >> 4);               // key tiles of 16 (Nk % 16 == 0, <= 160)
        const a16_t *kb = (const a16_t *)p.k + (int64_t)img * p.Nk * p.k_rs;
        const a16_t *vb = (const a16_t *)p.vt + (int64_t)img * p.vt_bs;
        uint4 kreg[4], vreg[4];
        auto load_head = [&](int hd) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 256 * i;                  // chunk c: key c / 5, 16-byte piece c % 5 of the head's 80 B (800 chunks:
                const int key = (c * 13108) >> 16, ch = c - key * 5;      // 160 rows; rows >= Nk are staged as zeros and masked in the softmax)
                kreg[i] = key < p.Nk ? *(const uint4 *)(kb + (int64_t)key * p.k_rs + hd * XD + ch * 8) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 256 * i;                  // chunk c: row c / 20 of V_h^T, keys 8 (c % 20) ..
                const int row = (c * 3277) >> 16, ch = c - row * 20;
                vreg[i] = (row < XD && ch * 8 < p.Nk_pad) ? *(const uint4 *)(vb + (int64_t)(hd * XD + row) * p.Nk_pad + ch * 8)
                                                          : make_uint4(0, 0, 0, 0);
            }
        };
        auto store_head = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 256 * i;
                const int key = (c * 13108) >> 16, ch = c - key * 5;
                if (key < 160) *(uint4 *)(smem + X_KIMG + key * X_KROW + ch * 16) = kreg[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 256 * i;
                const int row = (c * 3277) >> 16, ch = c - row * 20;
                if (row < XD) *(uint4 *)(smem + X_VIMG + row * X_VROW + ch * 16) = vreg[i];
            }
        };
        load_head(0);
        __syncthreads();                                      // q is complete in X; the ring is idle
        XA_TL(3);
        store_head();
        const int px = 16 * wave + r16;
        const int sw = ((px >> 2) & 1) << 1;
#pragma unroll 1
        for (int hd = 0; hd < XHEADS; ++hd) {
            __syncthreads();                                  // K_h / V_h^T of this head are in LDS
            if (hd + 1 < XHEADS) load_head(hd + 1);
            // q_h as the B operand: dims 8 g .. 8 g + 7 (K-step 0), dims 32 .. 39 on lane group 0 (K-step 1, zeros elsewhere)
            const int G0 = 5 * hd + g, G1 = 5 * hd + 4;
            const a16x8 qf0 = *(const a16x8 *)(smem + (G0 >> 2) * (XM * 64) + px * 64 + (((G0 & 3) ^ sw) << 4));
            a16x8 qf1 = *(const a16x8 *)(smem + (G1 >> 2) * (XM * 64) + px * 64 + (((G1 & 3) ^ sw) << 4));
            if (g != 0) qf1 = a16x8{0, 0, 0, 0, 0, 0, 0, 0};
            f32x4 s[10];
            a16x8 kf0[10], kf1[10];
#pragma unroll
            for (int kt = 0; kt < 10; ++kt) {
                const char *kp = smem + X_KIMG + (16 * kt + r16) * X_KROW;
                kf0[kt] = *(const a16x8 *)(kp + g * 16);
                kf1[kt] = *(const a16x8 *)(kp + 64);                          // every lane group the same finite granule: times zero for g != 0
            }
            float mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 10; ++kt) {
                if (KT_ && kt >= KT_) {                       // (compile-time: the tile does not exist)
                    s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    continue;
                }
                XA_MFMA_Z(s[kt], kf0[kt], qf0);               // C = 0 as an inline constant: no zeroed accumulator registers
                s[kt] = ONEDC_MFMA16(kf1[kt], qf1, s[kt]);
            }
#pragma unroll
            for (int kt = 0; kt < 10; ++kt) {
                const float t = fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3]));
                mx = kt < KT ? fmaxf(mx, t) : mx;                             // (keys >= Nk: zero rows of the image, masked here and below)
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m = mx * scale_log2e;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 10; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = kt < KT ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][e], scale_log2e, -m)) : 0.f;
                    s[kt][e] = pv;
                    sum += pv;
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            // O^T += V_h^T . P^T: K-step st covers key tiles 2 st, 2 st + 1 -- element j < 4 of a lane's fragment is key 32 st + 4 g + j,
            // element j >= 4 key 32 st + 16 + 4 g + (j - 4): the accumulator registers as they stand (attention.hip)
            f32x4 o[3];
#pragma unroll
            for (int st = 0; st < 5; ++st) {
                union { a16x8 v; uint32_t u[4]; } pk;
                if (VT_BF16) {
                    pk.u[0] = pack_bf16_2(s[2 * st][0], s[2 * st][1]);
                    pk.u[1] = pack_bf16_2(s[2 * st][2], s[2 * st][3]);
                    pk.u[2] = pack_bf16_2(s[2 * st + 1][0], s[2 * st + 1][1]);
                    pk.u[3] = pack_bf16_2(s[2 * st + 1][2], s[2 * st + 1][3]);
                } else {
                    pk.u[0] = pack_a2(s[2 * st][0], s[2 * st][1]);
                    pk.u[1] = pack_a2(s[2 * st][2], s[2 * st][3]);
                    pk.u[2] = pack_a2(s[2 * st + 1][0], s[2 * st + 1][1]);
                    pk.u[3] = pack_a2(s[2 * st + 1][2], s[2 * st + 1][3]);
                }
                {
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt) {
                        const int row = 16 * dt + r16;                     // dims 40 .. 47 of the third tile: any real row (results unused)
                        const char *vp = smem + X_VIMG + (row < XD ? row : row - 8) * X_VROW + (32 * st + 4 * g) * 2;
                        union { a16x8 v; uint2 h2[2]; } vf;
                        vf.h2[0] = *(const uint2 *)vp;
                        vf.h2[1] = *(const uint2 *)(vp + 32);
                        if (st == 0) {
                            if (VT_BF16) XA_MFMA_Z_BF16(o[dt], vf.v, pk.v);
                            else XA_MFMA_Z(o[dt], vf.v, pk.v);
                        } else if (VT_BF16) {
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pk.v, o[dt], 0, 0, 0);
                        } else {
                            o[dt] = ONEDC_MFMA16(vf.v, pk.v, o[dt]);
                        }
                    }
                }
            }
            const float inv = 1.f / sum;
            // o_h -> X in place of q_h (this wave's pixels only: q_h is in registers, the other heads' granules are untouched)
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int d0 = 16 * dt + 4 * g;
                if (d0 < XD) {
                    const int c = XD * hd + d0;
                    *(uint2 *)(smem + x_addr(c & ~7, px) + ((c >> 2) & 1) * 8) =
                        make_uint2(pack_a2(o[dt][0] * inv, o[dt][1] * inv), pack_a2(o[dt][2] * inv, o[dt][3] * inv));
                }
            }
            __syncthreads();                                  // every wave is done with this head's K / V^T images
            if (hd + 1 < XHEADS) store_head();
        }
    }

    XA_TL(4);
    // ---- P4: y = Wo . o + bias + h ---------------------------------------------------------------------------------------------------
    // (the last barrier of the head loop: o is complete in X and the ring region is free again)
    uint2 res[5][4];
    {
        const a16_t *hp = (const a16_t *)p.h + (row0 + r16) * p.h_cs + p.h_co + 80 * wave + 4 * g;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) res[i][pt] = *(const uint2 *)(hp + (int64_t)(16 * pt) * p.h_cs + 16 * i);
    }
    gemm_issue(p.wo);
    gemm_run();
    __syncthreads();                                          // every wave is past its last read of X and of its ring slots
    XA_TL(5);

    // ---- P5: bias, residual, 16-bit, through LDS (pixel rows of 640 B + 16), whole rows out ------------------------------------------
    {
        float4 bias[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) bias[i] = p.bo ? *(const float4 *)(p.bo + 80 * wave + 16 * i + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const f32x4 a = acc[i][pt];
                const float v0 = a[0] + bias[i].x + a2f_lo(res[i][pt].x), v1 = a[1] + bias[i].y + a2f_hi(res[i][pt].x);
                const float v2 = a[2] + bias[i].z + a2f_lo(res[i][pt].y), v3 = a[3] + bias[i].w + a2f_hi(res[i][pt].y);
                *(uint2 *)(smem + (16 * pt + r16) * X_SROW + (80 * wave + 16 * i + 4 * g) * 2) = make_uint2(pack_a2(v0, v1), pack_a2(v2, v3));
            }
        __syncthreads();
        a16_t *yb = (a16_t *)p.y + (row0 + 16 * wave) * p.y_cs + p.y_co;
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            const int idx = it * 64 + lane;                   // 16 pixels x 40 pieces of 16 B
            const int pl = (idx * 1639) >> 16, ch = idx - pl * 40;
            const uint4 o = *(const uint4 *)(smem + (16 * wave + pl) * X_SROW + ch * 16);
            *(uint4 *)(yb + (int64_t)pl * p.y_cs + ch * 8) = o;
        }
#ifdef EXP_XA_TIMELINE
        XA_TL(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        XA_TL(7);
        __syncthreads();
        if (tid == 0) {
            unsigned long long *dbg = (unsigned long long *)((a16_t *)p.y + row0 * p.y_cs + p.y_co);
#pragma unroll
            for (int i = 0; i < 8; ++i) dbg[i] = tl[i];
        }
#endif
    }
}

}  // namespace

extern "C" {

// 1 where onedc_xattn_block takes the shape (the host keeps the four separate launches otherwise)
int onedc_xattn_block_ok(const onedc_xattn_params *p) {
    if (!p || !p->h || (!p->ln_g != !p->ln_b) || !p->wq || !p->k || !p->vt || !p->wo || !p->y) return 0;
    if (p->C != XC || p->heads != XHEADS || p->d != XD) return 0;
    if (p->Nk < 16 || p->Nk > 160 || (p->Nk & 15) || p->Nk_pad < p->Nk || (p->Nk_pad & 7)) return 0;
    if (p->HW <= 0 || (p->HW % XM) || p->B <= 0 || (int64_t)p->B * p->HW / XM > 0x7fffffff) return 0;
    if (((p->h_cs | p->h_co | p->y_cs | p->y_co | p->k_rs) & 7) || p->k_rs < XHEADS * XD) return 0;
    if (p->h_co < 0 || p->y_co < 0 || p->h_co + XC > p->h_cs || p->y_co + XC > p->y_cs) return 0;      // the channel slices lie inside their rows
    if (p->vt_bs < (int64_t)XHEADS * XD * p->Nk_pad || p->h == p->y) return 0;
    return 1;
}

int onedc_xattn_block(const onedc_xattn_params *p, void *stream) {
    if (!onedc_xattn_block_ok(p)) return -1;
    static OnedcLdsAttrOnce once;
    once.ensure({{(const void *)xattn_block_kernel<true, 9>, X_LDS}, {(const void *)xattn_block_kernel<false, 9>, X_LDS},
                 {(const void *)xattn_block_kernel<true, 0>, X_LDS}, {(const void *)xattn_block_kernel<false, 0>, X_LDS}});
    const dim3 grid((unsigned)((int64_t)p->B * p->HW / XM)), block(256);
    const float scale_log2e = p->scale * 1.4426950408889634f;
#define XA_LAUNCH(BF_, KT_) hipLaunchKernelGGL((xattn_block_kernel<BF_, KT_>), grid, block, X_LDS, as_stream(stream), *p, scale_log2e)
    if (p->Nk == 144) {                                       // the 768 x 768 picture's context: the key-tile count at compile time
        if (p->vt_bf16) XA_LAUNCH(true, 9);
        else XA_LAUNCH(false, 9);
    } else {
        if (p->vt_bf16) XA_LAUNCH(true, 0);
        else XA_LAUNCH(false, 0);
    }
#undef XA_LAUNCH
    ONEDC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
