// attention.hip — flash attention for the UNet's multi-head self / cross attention.
//
// One workgroup = 4 wave64s = 128 queries of one (batch, head); each wave owns 32 queries.
// Everything is computed TRANSPOSED so the softmax is lane-local:
//   S^T = K . Q^T     (MFMA A = K tile rows from LDS, B = Q fragments held in registers)
//   O^T += V^T . P^T  (MFMA A = V^T tile rows from LDS, B = P^T built from the S^T accumulators
//                      in registers: with the query on the lane, the accumulator of one product is
//                      already the B operand of the next, in a permuted but consistent key order)
// A lane owns one query column: its running max / rescale never leave the lane except for a
// two-step butterfly over the four lane quads that split the keys.
//
// With head dims of 40/80 this kernel is VALU-bound, not MFMA-bound (4*d MFMA flops per
// (query, key) pair against a fixed softmax cost), so the softmax is kept to ~3.5 VALU ops per pair:
//   * p = exp2(s*c - m) is ONE fma + ONE v_exp_f32 (c = scale*log2 e, m tracked in the scaled domain)
//   * the row sum rides on the PV MFMA: where the padded head has a spare V^T row (d = 40 -> 48,
//     80 -> 96) that row is staged as all-ones, so O^T[d] accumulates sum(p) with the same rescales
//   * key masking only exists in the tail-tile instantiation; the O rescale is skipped while the
//     running max of the whole wave is unchanged; bf16 packing is v_cvt_pk_bf16_f32
// K/V^T tiles of 64 keys are register-staged into padded LDS rows (the guide's recommendation for
// attention), double buffered: the loads of tile t+1 are issued before the math of tile t and written
// to the other buffer after it, one barrier per tile.  V arrives transposed ([channel][key]) from
// the projection GEMM, which produces it in that layout for free by swapping its operands.
#include <type_traits>

#include "common.h"

namespace {

template <int DQ, int DV, bool ONES_ROW>
struct FA {
    static constexpr int KROW = DQ * 2 + 16;     // padded LDS row bytes of the K tile
    static constexpr int VROW = 64 * 2 + 16;     // padded LDS row bytes of the V^T tile (64 keys)
    static constexpr int QS = DQ / 32;           // k-steps of the S product
    static constexpr int DT = DV / 16;           // output tiles along d
    static constexpr int KCH = DQ / 8;           // 16-byte chunks per K row
    static constexpr int KCHUNKS = 64 * KCH, VCHUNKS = DV * 8;
    static constexpr int NKL = (KCHUNKS + 255) / 256, NVL = (VCHUNKS + 255) / 256;
    static constexpr int STAGE = 64 * KROW + DV * VROW;
};

template <int DQ, int DV, bool ONES_ROW, bool PREFETCH>
__global__ __launch_bounds__(256) void flash_attn_kernel(const a16_t *__restrict__ Q,
                                                         const a16_t *__restrict__ K,
                                                         const a16_t *__restrict__ VT,
                                                         a16_t *__restrict__ O, int H, int Nq, int Nk,
                                                         int Nk_pad, float scale_log2e, int d_real, int q_rs, int k_rs,
                                                              int64_t vt_bs) {
    using C = FA<DQ, DV, ONES_ROW>;
    constexpr int NBUF = PREFETCH ? 2 : 1;
    __shared__ __attribute__((aligned(16))) char lds[NBUF * C::STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;

    // Q fragments (B operand): lane (g, r) holds Q[q0 + 16*t + r][32*ks + 8*g .. +7]
    a16x8 qf[2][C::QS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = q0 + t * 16 + r;
#pragma unroll
        for (int ks = 0; ks < C::QS; ++ks) {
            if (qi < Nq)
                qf[t][ks] = *(const a16x8 *)(Q + ((int64_t)b * Nq + qi) * q_rs + h * DQ + ks * 32 + g * 8);
            else
                qf[t][ks] = a16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    f32x4 o[C::DT][2];
#pragma unroll
    for (int d = 0; d < C::DT; ++d)
#pragma unroll
        for (int t = 0; t < 2; ++t) o[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

    const a16_t *Kb = K + (int64_t)b * Nk * k_rs + h * DQ;
    const a16_t *Vb = VT + (int64_t)b * vt_bs + (int64_t)h * DV * Nk_pad;
    const uint4 ones = make_uint4(ONEDC_A16_ONE2, ONEDC_A16_ONE2, ONEDC_A16_ONE2, ONEDC_A16_ONE2);

    uint4 kreg[C::NKL], vreg[C::NVL];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < C::NKL; ++i) {
            const int c = tid + 256 * i;
            const int row = c / C::KCH, ch = c - row * C::KCH;
            kreg[i] = make_uint4(0, 0, 0, 0);
            if ((C::KCHUNKS % 256 == 0 || c < C::KCHUNKS) && k0 + row < Nk)
                kreg[i] = *(const uint4 *)(Kb + (int64_t)(k0 + row) * k_rs + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < C::NVL; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 3, ch = c & 7;
            if (C::VCHUNKS % 256 == 0 || c < C::VCHUNKS) {
                vreg[i] = *(const uint4 *)(Vb + (int64_t)row * Nk_pad + k0 + ch * 8);
                if (ONES_ROW && row == d_real) vreg[i] = ones;
            }
        }
    };
    auto store_tile = [&](char *buf) {
#pragma unroll
        for (int i = 0; i < C::NKL; ++i) {
            const int c = tid + 256 * i;
            const int row = c / C::KCH, ch = c - row * C::KCH;
            if (C::KCHUNKS % 256 == 0 || c < C::KCHUNKS) *(uint4 *)(buf + row * C::KROW + ch * 16) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < C::NVL; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 3, ch = c & 7;
            if (C::VCHUNKS % 256 == 0 || c < C::VCHUNKS)
                *(uint4 *)(buf + 64 * C::KROW + row * C::VROW + ch * 16) = vreg[i];
        }
    };

    auto tile_math = [&](const char *buf, int k0, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        const char *ldsK = buf;
        const char *ldsV = buf + 64 * C::KROW;
        // ---- S^T tiles: s[kt][t] = keys 16*kt..+15 (rows) x queries of tile t (cols)
        f32x4 s[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int t = 0; t < 2; ++t) s[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C::QS; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const a16x8 kf = *(const a16x8 *)(ldsK + (kt * 16 + r) * C::KROW + (ks * 4 + g) * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    s[kt][t] = ONEDC_MFMA16(kf, qf[t][ks], s[kt][t]);
            }
        }
        // ---- online softmax, lane-local per query column
        a16x8 pf[2][2];   // [t][step]: P^T B-operand fragments
        float alpha[2];
        bool changed = false;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (MASK) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k0 + kt * 16 + g * 4 + e >= Nk) s[kt][t][e] = -1e30f;
            }
            float mx = fmaxf(fmaxf(s[0][t][0], s[0][t][1]), fmaxf(s[0][t][2], s[0][t][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt)
                mx = fmaxf(fmaxf(mx, fmaxf(s[kt][t][0], s[kt][t][1])), fmaxf(s[kt][t][2], s[kt][t][3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[t], mx * scale_log2e);
            alpha[t] = __builtin_amdgcn_exp2f(m_run[t] - m_new);
            changed |= m_new != m_run[t];
            m_run[t] = m_new;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][t][e], scale_log2e, -m_new));
                    s[kt][t][e] = pv;
                    if (!ONES_ROW) sum += pv;
                }
            if (!ONES_ROW) l_run[t] = l_run[t] * alpha[t] + sum;
            // step st covers key tiles 2*st and 2*st+1: element j<4 -> tile 2*st, j>=4 -> tile 2*st+1
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                union { a16x8 v; uint32_t u[4]; } pk;
                pk.u[0] = pack_a2(s[2 * st][t][0], s[2 * st][t][1]);
                pk.u[1] = pack_a2(s[2 * st][t][2], s[2 * st][t][3]);
                pk.u[2] = pack_a2(s[2 * st + 1][t][0], s[2 * st + 1][t][1]);
                pk.u[3] = pack_a2(s[2 * st + 1][t][2], s[2 * st + 1][t][3]);
                pf[t][st] = pk.v;
            }
        }
        if (__any(changed)) {                           // wave-uniform: skipped once the max has settled
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int d = 0; d < C::DT; ++d)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[d][t][e] *= alpha[t];
        }
        // ---- O^T += V^T . P^T ; A fragment of lane (r, g): V^T[16*d + r][keys 32*st + 4g + (0..3)
        //      and 32*st + 16 + 4g + (0..3)] -- the same permuted key order as pf
#pragma unroll
        for (int d = 0; d < C::DT; ++d) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const char *vp = ldsV + (d * 16 + r) * C::VROW + (st * 32 + g * 4) * 2;
                union { a16x8 v; uint2 h2[2]; } vf;
                vf.h2[0] = *(const uint2 *)vp;
                vf.h2[1] = *(const uint2 *)(vp + 32);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    o[d][t] = ONEDC_MFMA16(vf.v, pf[t][st], o[d][t]);
            }
        }
    };

    const int ntiles = (Nk + 63) >> 6;
    const bool ragged = (Nk & 63) != 0;
    if (PREFETCH) {
        load_tile(0);
        store_tile(lds);
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            const char *cur = lds + (t & 1) * C::STAGE;
            if (t + 1 < ntiles) load_tile((t + 1) * 64);
            if (ragged && t == ntiles - 1) tile_math(cur, t * 64, std::true_type{});
            else tile_math(cur, t * 64, std::false_type{});
            if (t + 1 < ntiles) store_tile(lds + ((t + 1) & 1) * C::STAGE);
            __syncthreads();
        }
    } else {
        for (int t = 0; t < ntiles; ++t) {
            __syncthreads();
            load_tile(t * 64);
            store_tile(lds);
            __syncthreads();
            if (ragged && t == ntiles - 1) tile_math(lds, t * 64, std::true_type{});
            else tile_math(lds, t * 64, std::false_type{});
        }
    }

    // ---- normalise and store: lane (g, r) owns channels 16*d + 4g + (0..3) of query q0 + 16*t + r
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float l;
        if (ONES_ROW) {
            // sum(p) sits in O^T[d_real][q]: element (d_real & 3) of tile d_real/16 in lane quad (d_real/4)&3
            const int dt = d_real >> 4, dq = (d_real >> 2) & 3, de = d_real & 3;
            float cand = 0.f;
#pragma unroll
            for (int d = 0; d < C::DT; ++d)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (d == dt && e == de) cand = o[d][t][e];
            l = __shfl(cand, dq * 16 + r);
        } else {
            l = l_run[t];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        const float inv = 1.f / l;
        const int qi = q0 + t * 16 + r;
        if (qi >= Nq) continue;
        a16_t *op = O + ((int64_t)b * Nq + qi) * (H * DV) + h * DV + g * 4;
#pragma unroll
        for (int d = 0; d < C::DT; ++d) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = o[d][t][e] * inv;
                if (ONES_ROW && d * 16 + g * 4 + e >= d_real) v[e] = 0.f;   // keep the pad channels zero
            }
            *(uint2 *)(op + d * 16) = make_uint2(pack_a2(v[0], v[1]), pack_a2(v[2], v[3]));
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// flash_attn64_kernel: head dim 40 (DQ = 64, DV = 48), the shape that carries 80 % of the UNet's
// attention time (self-attention over 9216 tokens).  Same transposed formulation and softmax as above;
// what changes is the machine mapping:
//   * a wave owns 64 queries (4 MFMA column tiles): every K / V^T fragment read from LDS feeds 4 MFMAs
//   * K (64 keys x 128 B) and V^T (48 rows x 128 B) tiles are 128-byte-row tiles, streamed by LDS-DMA
//     (global_load_lds_dwordx4, no VGPR round trip) into a 3-deep ring, counted vmcnt, one raw
//     s_barrier per tile; 16-byte chunk c of row r sits at c ^ (r & 7) for V^T and at
//     c ^ ((r & 3) | ((r >> 3 & 1) << 2)) for K (applied on the DMA source address and on the reads)
//   * the S^T product reads K rows in the order key(tile h, row i) = 32 st + 8 (i >> 2) + 4 h + (i & 3):
//     the accumulator registers of a lane are then 8 CONSECUTIVE keys per 32-key MFMA step, so the V^T
//     A-fragments are single ds_read_b128 (conflict-free), no split 8-byte reads
//   * the all-ones V^T row that makes the PV product accumulate the softmax denominator comes from a
//     constant page: the DMA source of that LDS row is redirected, V^T itself is read-only
//   * 4 waves / 42 KB per workgroup -> two workgroups per CU drift apart, so one SIMD overlaps the exp
//     phase of one wave with the MFMA phase of another
// (bf16 ones in either build: V^T and P of this kernel are bf16)
__device__ const uint32_t g_ones_page[64] = {
    0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u,
    0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u,
    0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u,
    0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u,
    0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u,
    0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u,
    0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u,
    0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

// bf16 {1, 0, 0, 0, 0, 0, 0, 0}: stands in for chunk 5 (the all-padding dims 40..47) of every K row, so
// that dim 40 of the S product is a free "+ 1 * q[40]" term: q[40] carries -m_run
__device__ const uint32_t g_kone_page[4] = {ONEDC_A16_ONE, 0u, 0u, 0u};
// COMPACT heads (round 5): Q, K, V^T and O hold the d REAL dims of a head back to back in memory (head stride d = 40 / 80 elements
// instead of the padded 64 / 48 / 128 / 96); the padding only exists in LDS and in registers: the 16-byte chunks of a K row and
// the V^T rows that are padding are DMA'd from this page of zeros, like the constant-1 chunk and the ones row already are.  The
// projections then compute and move no padded columns (to_q|k 320 -> 640 instead of -> 1024, to_out K = 320 instead of 384), and
// the values every MFMA sees are the same as with padded tensors: bit-identical results.
__device__ const uint32_t g_zero_page[64] = {0u};

constexpr int FA64_K = 64 * 128;                 // K tile bytes
constexpr int FA64_V = 48 * 128;                 // V^T tile bytes
constexpr int FA64_STAGE = FA64_K + FA64_V;      // 14 KB
constexpr int FA64_NST = 3;

__device__ __forceinline__ int fa_xcd_logical(int bid, int nblk) {      // bijective, see conv_common.h
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
}

__device__ __forceinline__ void fa_glds16(const char *src, char *lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// FA64_QT query column tiles (16 queries each) per wave, FA64_OCC workgroups per CU.  4 / 2 is the product; 3 / 3 (48 queries per
// wave, 168 registers, three waves per SIMD) is round 5's experiment: -DFA64_QT=3 -DFA64_OCC=3 (tools/exp_build.sh)
#ifndef FA64_QT
#define FA64_QT 4
#define FA64_OCC 2
#endif
template <bool COMPACT>
__global__ __launch_bounds__(256, FA64_OCC) void flash_attn64_kernel(const a16_t *__restrict__ Q,
                                                              const a16_t *__restrict__ K,
                                                              const a16_t *__restrict__ VT,
                                                              a16_t *__restrict__ O, int H, int Nq, int Nk,
                                                              int Nk_pad, float scale_log2e, int d_real, int q_rs, int k_rs,
                                                              int64_t vt_bs) {
    constexpr int DQ = 64, DV = 48, QT = FA64_QT, QW = 64 * QT;      // QW: queries per workgroup
    constexpr int HQ = COMPACT ? 40 : DQ, HV = COMPACT ? 40 : DV;    // head strides in memory (elements / V^T rows)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    // all query blocks of one (image, head) run on ONE XCD, back to back: its K / V^T (2 MB at 9216 keys)
    // stay in that XCD's L2 instead of being pulled through all eight
    const int nqb = (Nq + QW - 1) / QW;
    const int logical = fa_xcd_logical(blockIdx.x, gridDim.x);
    const int bh = logical / nqb;
    const int h = bh % H, b = bh / H;
    const int q0 = (logical - bh * nqb) * QW + wave * 16 * QT;

    // Q fragments (B operand): lane (g, r) holds Q[q0 + 16 t + r][32 ks + 8 g .. + 7], pre-multiplied by
    // scale * log2(e): the S product then lands in the exp2 domain and needs no per-element multiply
    a16x8 qf[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + t * 16 + r;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { a16x8 v; uint32_t u[4]; } f;
            f.v = a16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (qi < Nq && (!COMPACT || ks * 4 + g < 5))
                f.v = *(const a16x8 *)(Q + ((int64_t)b * Nq + qi) * q_rs + h * HQ + ks * 32 + g * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                f.u[e] = pack_a2(a2f((a16_t)(f.u[e] & 0xffff)) * scale_log2e, a2f((a16_t)(f.u[e] >> 16)) * scale_log2e);
            qf[t][ks] = f.v;
        }
    }
    f32x4 o[3][QT];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int t = 0; t < QT; ++t) o[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QT];                                // running maximum (exp2 domain), identical in the 4 lanes of a query
#pragma unroll
    for (int t = 0; t < QT; ++t) m_run[t] = 0.f;

    // ---- DMA geometry: 14 blocks of 8 LDS rows per tile: K blocks 0..7 (two per wave), V^T blocks 0..5
    // (waves 0, 1: two; waves 2, 3: one).  lane -> (row in block, 16-byte chunk)
    const int lrow = lane >> 3, lch = lane & 7;
    const char *Kb = (const char *)(K + (int64_t)b * Nk * k_rs + h * HQ);
    const char *Vb = (const char *)(VT + (int64_t)b * vt_bs + (int64_t)h * HV * Nk_pad);
    const int64_t krow_bytes = (int64_t)k_rs * 2;
    const int64_t vrow_bytes = (int64_t)Nk_pad * 2;
    int krow[2], kchunk[2];
    const char *kfix[2];                              // the constant source of this lane's chunk (nullptr: the K row itself)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        krow[i] = (wave + 4 * i) * 8 + lrow;                                  // key row inside the tile
        kchunk[i] = lch ^ ((krow[i] & 3) | (((krow[i] >> 3) & 1) << 2));
        kfix[i] = kchunk[i] == 5 ? (const char *)g_kone_page : (COMPACT && kchunk[i] > 5) ? (const char *)g_zero_page : nullptr;
    }
    const int nvb = wave < 2 ? 2 : 1;                                         // V^T blocks of this wave
    int vrow[2];
    const char *vsrc[2];
    bool vones[2];                                    // (constant source: the ones row; COMPACT: the padding rows too)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        vrow[i] = (wave + 4 * i) * 8 + lrow;                                  // < 48 for the blocks that exist
        const int ch = lch ^ (vrow[i] & 7);
        vones[i] = vrow[i] == d_real || (COMPACT && vrow[i] > d_real);
        vsrc[i] = vrow[i] == d_real ? (const char *)g_ones_page + ch * 16
                  : (COMPACT && vrow[i] > d_real) ? (const char *)g_zero_page : Vb + (int64_t)vrow[i] * vrow_bytes + ch * 16;
    }
    auto issue = [&](int tile, char *stage) {
        const int k0 = tile * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int key = k0 + krow[i];
            key = key < Nk ? key : Nk - 1;                                    // tail rows: any valid row, masked below
            fa_glds16(kfix[i] ? kfix[i] : Kb + (int64_t)key * krow_bytes + kchunk[i] * 16, stage + (wave + 4 * i) * 1024);
        }
        fa_glds16(vsrc[0] + (vones[0] ? 0 : (int64_t)k0 * 2), stage + FA64_K + wave * 1024);
        if (nvb == 2) fa_glds16(vsrc[1] + (vones[1] ? 0 : (int64_t)k0 * 2), stage + FA64_K + (wave + 4) * 1024);
    };

    // ---- fragment read offsets
    // K A-fragment of (st, hh, ks): LDS row 32 st + 8 (r >> 2) + 4 hh + (r & 3), chunk (4 ks + g) ^ swzK(row)
    int kfo[2][2];                                  // [hh][ks] for st = 0; st = 1 adds 32 rows (same swizzle bits)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = 8 * (r >> 2) + 4 * hh + (r & 3);
        const int sw = (row & 3) | (((row >> 3) & 1) << 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kfo[hh][ks] = row * 128 + (((4 * ks + g) ^ sw) << 4);
    }
    // V^T A-fragment of (d, st): LDS row 16 d + r, chunk (4 st + g) ^ (row & 7)
    int vfo[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) vfo[st] = r * 128 + (((4 * st + g) ^ (r & 7)) << 4);

    // Softmax with a lazily moved reference point.  Dim 40 (head padding) of every K row reads as 1 and
    // dim 40 of Q' holds -m_run (kept bf16-representable), so the MFMA chain itself delivers s - m_run and
    // p = exp2(s - m_run) is ONE v_exp_f32 per element: no multiply, no subtract, no cross-lane traffic.
    // m_run only has to keep exp2 in range (bf16 and f32 share the 8-bit exponent, and the relative
    // precision of p does not depend on its magnitude), so it is set from the row maxima of tile 0 and
    // moved only when a later tile holds a score more than 2^FA64_THR above it; the move is applied AFTER
    // that tile's PV product (O and the riding row sum are scaled together), so no register between the
    // S product and the PV product is ever touched by the rare path.
    // P and V^T are ALWAYS bf16 in this kernel, also in the IEEE-half build (Q, K and the S product are the build's
    // format): exp2 of a score that jumped far above the lazily moved reference is only representable with f32's
    // exponent range, and a pre-check in front of the PV product costs registers the loop does not have (it spilled:
    // 4.2 instead of 2.7 ms at 9216 tokens).  The host hands V^T over in bf16 (onedc_conv_params.out_bf16).
    constexpr float FA64_THR = 12.f;
    auto s_chain = [&](const char *ldsK, f32x4 (&s)[4][QT]) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int t = 0; t < QT; ++t) s[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const a16x8 kf = *(const a16x8 *)(ldsK + (kt >> 1) * 4096 + kfo[kt & 1][ks]);
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    s[kt][t] = ONEDC_MFMA16(kf, qf[t][ks], s[kt][t]);
            }
    };
    auto mask_tail = [&](f32x4 (&s)[4][QT], int k0) {      // element (kt, e) is key k0 + 32 (kt >> 1) + 8 g + 4 (kt & 1) + e
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + e >= Nk) {
#pragma unroll
                    for (int t = 0; t < QT; ++t) s[kt][t][e] = -1e30f;
                }
    };
    auto lane_max = [&](const f32x4 (&s)[4][QT], float (&tmax)[QT]) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = fmaxf(fmaxf(s[0][t][0], s[0][t][1]), fmaxf(s[0][t][2], s[0][t][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt)
                mx = fmaxf(fmaxf(mx, fmaxf(s[kt][t][0], s[kt][t][1])), fmaxf(s[kt][t][2], s[kt][t][3]));
            tmax[t] = mx;
        }
    };
    // move the reference point of query tile t by the row maximum `mx` (relative to the current one)
    auto move_ref = [&](int t, float mx, bool scale_o) -> float {
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = a2f(f2a(m_run[t] + mx));              // bf16 grid: it must fit Q'[40] exactly
        const float inc = m_new - m_run[t];                          // exact (both on the bf16 grid)
        m_run[t] = m_new;
        if (g == 1) {                                                // dim 40 = element 0 of the (ks = 1, g = 1) fragment
            union { a16x8 v; uint32_t u[4]; } f;
            f.v = qf[t][1];
            f.u[0] = (f.u[0] & 0xffff0000u) | f2a(-m_new);
            qf[t][1] = f.v;
        }
        if (scale_o) {
            const float alpha = __builtin_amdgcn_exp2f(-inc);
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[d][t][e] *= alpha;
        }
        return inc;
    };
    auto first_reference = [&](const char *stage, bool mask) {      // dry run of the S product on tile 0
        f32x4 s[4][QT];
        s_chain(stage, s);
        if (mask) mask_tail(s, 0);
        float tmax[QT];
        lane_max(s, tmax);
#pragma unroll
        for (int t = 0; t < QT; ++t) move_ref(t, tmax[t], false);
    };
    auto tile_math = [&](const char *stage, int k0, auto mask_tag, auto track_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        constexpr bool TRACK = decltype(track_tag)::value;
        const char *ldsK = stage, *ldsV = stage + FA64_K;
        f32x4 s[4][QT];                             // S^T tiles s[kt][t], kt = 2 st + hh
        s_chain(ldsK, s);
        if (MASK) mask_tail(s, k0);
        float tmax[QT];
        if constexpr (TRACK) lane_max(s, tmax);
        a16x8 pf[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[kt][t][e] = __builtin_amdgcn_exp2f(s[kt][t][e]);
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                union { a16x8 v; uint32_t u[4]; } pk;
                pk.u[0] = pack_bf16_2(s[2 * st][t][0], s[2 * st][t][1]);
                pk.u[1] = pack_bf16_2(s[2 * st][t][2], s[2 * st][t][3]);
                pk.u[2] = pack_bf16_2(s[2 * st + 1][t][0], s[2 * st + 1][t][1]);
                pk.u[3] = pack_bf16_2(s[2 * st + 1][t][2], s[2 * st + 1][t][3]);
                pf[t][st] = pk.v;
            }
        }
        // O^T += V^T . P^T: lane (r, g) of the A operand holds V^T[16 d + r][keys 32 st + 8 g .. + 7]
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const a16x8 vf = *(const a16x8 *)(ldsV + d * 2048 + vfo[st]);
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    o[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[t][st], o[d][t], 0, 0, 0);
            }
        if constexpr (TRACK) {
            float wmax = tmax[0];
#pragma unroll
            for (int t = 1; t < QT; ++t) wmax = fmaxf(wmax, tmax[t]);
            if (__any(wmax > FA64_THR)) {           // wave-uniform, rare
#pragma unroll
                for (int t = 0; t < QT; ++t) (void)move_ref(t, fmaxf(tmax[t], 0.f), true);
            }
        }
    };

    const int ntiles = (Nk + 63) >> 6;
    const bool ragged = (Nk & 63) != 0;
    auto pass = [&](auto track_tag) {
        issue(0, smem);
        if (ntiles > 1) issue(1, smem + FA64_STAGE);
        int slot = 0;
        for (int t = 0; t < ntiles; ++t) {
            // tile t has landed for this wave once at most the DMAs of tile t + 1 are outstanding
            if (t + 1 < ntiles) {
                if (nvb == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();           // every wave's part of tile t is in LDS; tile t - 1 is fully read
            if (t + 2 < ntiles) issue(t + 2, smem + (slot >= 1 ? slot - 1 : 2) * FA64_STAGE);
            const char *cur = smem + slot * FA64_STAGE;
            if (t == 0) first_reference(cur, ragged && ntiles == 1);
            if (ragged && t == ntiles - 1) tile_math(cur, t * 64, std::true_type{}, track_tag);
            else tile_math(cur, t * 64, std::false_type{}, track_tag);
            slot = slot == 2 ? 0 : slot + 1;
        }
    };
    const int dt = d_real >> 4, dq = (d_real >> 2) & 3, de = d_real & 3;
    auto denominator = [&](int t) -> float {         // the riding row sum: channel d_real of the query's output
        float cand = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (d == dt && e == de) cand = o[d][t][e];
        return __shfl(cand, dq * 16 + r);
    };
    // FAST PASS: the reference point of a query stays where tile 0 put it and no tile looks for its maximum (64 v_max per lane
    // and tile: 14 % of the kernel at 9 216 keys).  exp2 then overflows only where a later score lies 2^100 above tile 0's
    // best -- which the row sum reports by itself: a denominator that is not a finite number below 2^100 (O is bounded by
    // it times |v|) sends the WHOLE workgroup through the pass again, this time with the moving reference.
#ifdef EXP_FA64_ALWAYS_TRACK                         // (A/B builds) the kernel as it was: the moving reference in the only pass
    pass(std::true_type{});
    if (false)
#else
    pass(std::false_type{});
#endif
    {
        bool bad = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            // 2^-100 < l < 2^100 on the bit pattern (positive floats order like their bits; inf, NaN and negatives fall outside;
            // this file is compiled with -fno-honor-nans)
            const uint32_t lb = __builtin_bit_cast(uint32_t, denominator(t));
            bad |= !(lb > 0x0d800000u && lb < 0x71800000u);
        }
        if (__syncthreads_or(bad)) {                 // (a barrier too: every wave is done with the ring)
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                m_run[t] = 0.f;
                if (g == 1) {                        // dim 40 of Q' back to -0
                    union { a16x8 v; uint32_t u[4]; } f;
                    f.v = qf[t][1];
                    f.u[0] &= 0xffff0000u;
                    qf[t][1] = f.v;
                }
#pragma unroll
                for (int d = 0; d < 3; ++d) o[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            pass(std::true_type{});
        }
    }

    // ---- normalise and store: lane (g, r) owns channels 16 d + 4 g + (0..3) of query q0 + 16 t + r
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float l = denominator(t);
        const float inv = 1.f / l;
        const int qi = q0 + t * 16 + r;
        if (qi >= Nq) continue;
        a16_t *op = O + ((int64_t)b * Nq + qi) * (H * HV) + h * HV + g * 4;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = o[d][t][e] * inv;
                if (d * 16 + g * 4 + e >= d_real) v[e] = 0.f;
            }
            if (!COMPACT || d * 16 + g * 4 < 40)      // (COMPACT: only the real dims exist in memory; 40 is a multiple of 4)
                *(uint2 *)(op + d * 16) = make_uint2(pack_a2(v[0], v[1]), pack_a2(v[2], v[3]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// flash_attn80_kernel: head dim 80 as TWO half-heads of the d = 40 layout whose scores add before the softmax:
// Q / K rows are [dims 0..39 | 24 pad | dims 40..79 | 24 pad] (DQ = 128), V^T rows [0..39 | ones row 40 | pad | 40..79 |
// pad] (DV = 96).  Every LDS image, swizzle, fragment offset and the softmax (reference point in dim 40 of the first
// half, denominator on the ones row) are flash_attn64_kernel's; what changes is the machine mapping: a K tile is two
// 8 KB half tiles (the S chain runs four k steps), a V^T tile two 6 KB half tiles (six output row tiles), a wave owns
// 32 queries (the accumulators of 64 would not fit) and a workgroup is EIGHT waves sharing one 3 x 28 KB ring, one
// workgroup per CU: each K / V^T byte is staged once per 256 queries, as in the d = 40 kernel.
constexpr int FA80_STAGE = 2 * FA64_K + 2 * FA64_V;      // 28 KB

template <bool COMPACT>
__global__ __launch_bounds__(512, 2) void flash_attn80_kernel(const a16_t *__restrict__ Q,
                                                              const a16_t *__restrict__ K,
                                                              const a16_t *__restrict__ VT,
                                                              a16_t *__restrict__ O, int H, int Nq, int Nk,
                                                              int Nk_pad, float scale_log2e, int d_real, int q_rs, int k_rs,
                                                              int64_t vt_bs) {
    constexpr int DQ = 128, DV = 96, QT = 2;   // two half-heads of (64, 48): dims 0..39 | pad | 40..79 | pad
    constexpr int HQ = COMPACT ? 80 : DQ, HV = COMPACT ? 80 : DV;    // head strides in memory; COMPACT: dims 0..79 back to back
    constexpr int HALF_Q = COMPACT ? 40 : 64, HALF_V = COMPACT ? 40 : 48;   // where the second half-head starts
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    // all query blocks of one (image, head) run on ONE XCD, back to back: its K / V^T (2 MB at 9216 keys)
    // stay in that XCD's L2 instead of being pulled through all eight
    const int nqb = (Nq + 255) >> 8;                // 8 waves x 32 queries
    const int logical = fa_xcd_logical(blockIdx.x, gridDim.x);
    const int bh = logical / nqb;
    const int h = bh % H, b = bh / H;
    const int q0 = (logical - bh * nqb) * 256 + wave * 32;

    // Q fragments (B operand): lane (g, r) holds Q[q0 + 16 t + r][32 ks + 8 g .. + 7], pre-multiplied by
    // scale * log2(e): the S product then lands in the exp2 domain and needs no per-element multiply
    a16x8 qf[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + t * 16 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            union { a16x8 v; uint32_t u[4]; } f;
            f.v = a16x8{0, 0, 0, 0, 0, 0, 0, 0};
            // k step ks covers chunks 4 (ks & 1) + g of half-head ks >> 1
            if (qi < Nq && (!COMPACT || (ks & 1) * 4 + g < 5))
                f.v = *(const a16x8 *)(Q + ((int64_t)b * Nq + qi) * q_rs + h * HQ + (ks >> 1) * HALF_Q + (ks & 1) * 32 + g * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                f.u[e] = pack_a2(a2f((a16_t)(f.u[e] & 0xffff)) * scale_log2e, a2f((a16_t)(f.u[e] >> 16)) * scale_log2e);
            qf[t][ks] = f.v;
        }
    }
    f32x4 o[6][QT];                                 // o[3 hf + d]: rows 16 d + .. of half-head hf
#pragma unroll
    for (int d = 0; d < 6; ++d)
#pragma unroll
        for (int t = 0; t < QT; ++t) o[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QT];                                // running maximum (exp2 domain), identical in the 4 lanes of a query
#pragma unroll
    for (int t = 0; t < QT; ++t) m_run[t] = 0.f;

    // ---- DMA geometry: a stage is [K half 0 | K half 1 | V^T half 0 | V^T half 1] = 28 blocks of 8 LDS rows: wave w moves
    // K rows 8 w .. of both halves, V^T block w and (w < 4) block w + 8 (block vb: half vb / 6, rows 8 (vb % 6) ..)
    const int lrow = lane >> 3, lch = lane & 7;
    const char *Kb = (const char *)(K + (int64_t)b * Nk * k_rs + h * HQ);
    const char *Vb = (const char *)(VT + (int64_t)b * vt_bs + (int64_t)h * HV * Nk_pad);
    const int64_t krow_bytes = (int64_t)k_rs * 2;
    const int64_t vrow_bytes = (int64_t)Nk_pad * 2;
    const int krow = wave * 8 + lrow;                                         // key row inside the tile
    const int kchunk = lch ^ ((krow & 3) | (((krow >> 3) & 1) << 2));
    // constant sources of this lane's chunk in the two half tiles (nullptr: the K row itself)
    const char *const kfix0 = kchunk == 5 ? (const char *)g_kone_page : (COMPACT && kchunk > 5) ? (const char *)g_zero_page : nullptr;
    const char *const kfix1 = (COMPACT && kchunk >= 5) ? (const char *)g_zero_page : nullptr;
    const int nvb = wave < 4 ? 2 : 1;                                         // V^T blocks of this wave
    int vrow[2];
    const char *vsrc[2];
    bool vones[2];                                    // (constant source: the ones row; COMPACT: the padding rows too)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int vb = wave + 8 * i, hf = vb >= 6 ? 1 : 0;                    // (vb < 12 for the blocks that exist)
        vrow[i] = (vb - 6 * hf) * 8 + lrow;                                   // row inside the half: 0 .. 47
        const int ch = lch ^ (vrow[i] & 7);
        const bool ones = hf == 0 && vrow[i] == d_real;                       // the row that accumulates the denominator
        const bool pad = COMPACT && vrow[i] >= 40 && !ones;
        vones[i] = ones || pad;
        vsrc[i] = ones ? (const char *)g_ones_page + ch * 16 : pad ? (const char *)g_zero_page
                       : Vb + (int64_t)(hf * HALF_V + vrow[i]) * vrow_bytes + ch * 16;
    }
    auto issue = [&](int tile, char *stage) {
        const int k0 = tile * 64;
        int key = k0 + krow;
        key = key < Nk ? key : Nk - 1;                                        // tail rows: any valid row, masked below
        const char *ksrc = Kb + (int64_t)key * krow_bytes + kchunk * 16;
        fa_glds16(kfix0 ? kfix0 : ksrc, stage + wave * 1024);                                // half 0: dim 40 reads as 1
        fa_glds16(kfix1 ? kfix1 : ksrc + HALF_Q * 2, stage + FA64_K + wave * 1024);          // half 1: dims 40 .. 79
        fa_glds16(vsrc[0] + (vones[0] ? 0 : (int64_t)k0 * 2), stage + 2 * FA64_K + wave * 1024);
        if (nvb == 2) fa_glds16(vsrc[1] + (vones[1] ? 0 : (int64_t)k0 * 2), stage + 2 * FA64_K + (wave + 8) * 1024);
    };

    // ---- fragment read offsets
    // K A-fragment of (st, hh, ks): LDS row 32 st + 8 (r >> 2) + 4 hh + (r & 3), chunk (4 ks + g) ^ swzK(row)
    int kfo[2][2];                                  // [hh][ks] for st = 0; st = 1 adds 32 rows (same swizzle bits)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = 8 * (r >> 2) + 4 * hh + (r & 3);
        const int sw = (row & 3) | (((row >> 3) & 1) << 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kfo[hh][ks] = row * 128 + (((4 * ks + g) ^ sw) << 4);
    }
    // V^T A-fragment of (d, st): LDS row 16 d + r, chunk (4 st + g) ^ (row & 7)
    int vfo[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) vfo[st] = r * 128 + (((4 * st + g) ^ (r & 7)) << 4);

    // Softmax with a lazily moved reference point.  Dim 40 (head padding) of every K row reads as 1 and
    // dim 40 of Q' holds -m_run (kept bf16-representable), so the MFMA chain itself delivers s - m_run and
    // p = exp2(s - m_run) is ONE v_exp_f32 per element: no multiply, no subtract, no cross-lane traffic.
    // m_run only has to keep exp2 in range (bf16 and f32 share the 8-bit exponent, and the relative
    // precision of p does not depend on its magnitude), so it is set from the row maxima of tile 0 and
    // moved only when a later tile holds a score more than 2^FA64_THR above it; the move is applied AFTER
    // that tile's PV product (O and the riding row sum are scaled together), so no register between the
    // S product and the PV product is ever touched by the rare path.
    // P and V^T are ALWAYS bf16 in this kernel, also in the IEEE-half build (Q, K and the S product are the build's
    // format): exp2 of a score that jumped far above the lazily moved reference is only representable with f32's
    // exponent range, and a pre-check in front of the PV product costs registers the loop does not have (it spilled:
    // 4.2 instead of 2.7 ms at 9216 tokens).  The host hands V^T over in bf16 (onedc_conv_params.out_bf16).
    constexpr float FA64_THR = 12.f;
    auto s_chain = [&](const char *ldsK, f32x4 (&s)[4][QT]) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int t = 0; t < QT; ++t) s[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const a16x8 kf = *(const a16x8 *)(ldsK + (ks >> 1) * FA64_K + (kt >> 1) * 4096 + kfo[kt & 1][ks & 1]);
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    s[kt][t] = ONEDC_MFMA16(kf, qf[t][ks], s[kt][t]);
            }
    };
    auto mask_tail = [&](f32x4 (&s)[4][QT], int k0) {      // element (kt, e) is key k0 + 32 (kt >> 1) + 8 g + 4 (kt & 1) + e
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + e >= Nk) {
#pragma unroll
                    for (int t = 0; t < QT; ++t) s[kt][t][e] = -1e30f;
                }
    };
    auto lane_max = [&](const f32x4 (&s)[4][QT], float (&tmax)[QT]) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = fmaxf(fmaxf(s[0][t][0], s[0][t][1]), fmaxf(s[0][t][2], s[0][t][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt)
                mx = fmaxf(fmaxf(mx, fmaxf(s[kt][t][0], s[kt][t][1])), fmaxf(s[kt][t][2], s[kt][t][3]));
            tmax[t] = mx;
        }
    };
    // move the reference point of query tile t by the row maximum `mx` (relative to the current one)
    auto move_ref = [&](int t, float mx, bool scale_o) -> float {
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = a2f(f2a(m_run[t] + mx));              // bf16 grid: it must fit Q'[40] exactly
        const float inc = m_new - m_run[t];                          // exact (both on the bf16 grid)
        m_run[t] = m_new;
        if (g == 1) {                                                // dim 40 = element 0 of the (ks = 1, g = 1) fragment
            union { a16x8 v; uint32_t u[4]; } f;
            f.v = qf[t][1];
            f.u[0] = (f.u[0] & 0xffff0000u) | f2a(-m_new);
            qf[t][1] = f.v;
        }
        if (scale_o) {
            const float alpha = __builtin_amdgcn_exp2f(-inc);
#pragma unroll
            for (int d = 0; d < 6; ++d)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[d][t][e] *= alpha;
        }
        return inc;
    };
    auto first_reference = [&](const char *stage, bool mask) {      // dry run of the S product on tile 0
        f32x4 s[4][QT];
        s_chain(stage, s);
        if (mask) mask_tail(s, 0);
        float tmax[QT];
        lane_max(s, tmax);
#pragma unroll
        for (int t = 0; t < QT; ++t) move_ref(t, tmax[t], false);
    };
    auto tile_math = [&](const char *stage, int k0, auto mask_tag, auto track_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        constexpr bool TRACK = decltype(track_tag)::value;
        const char *ldsK = stage, *ldsV = stage + 2 * FA64_K;
        f32x4 s[4][QT];                             // S^T tiles s[kt][t], kt = 2 st + hh
        s_chain(ldsK, s);
        if (MASK) mask_tail(s, k0);
        float tmax[QT];
        if constexpr (TRACK) lane_max(s, tmax);
        a16x8 pf[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[kt][t][e] = __builtin_amdgcn_exp2f(s[kt][t][e]);
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                union { a16x8 v; uint32_t u[4]; } pk;
                pk.u[0] = pack_bf16_2(s[2 * st][t][0], s[2 * st][t][1]);
                pk.u[1] = pack_bf16_2(s[2 * st][t][2], s[2 * st][t][3]);
                pk.u[2] = pack_bf16_2(s[2 * st + 1][t][0], s[2 * st + 1][t][1]);
                pk.u[3] = pack_bf16_2(s[2 * st + 1][t][2], s[2 * st + 1][t][3]);
                pf[t][st] = pk.v;
            }
        }
        // O^T += V^T . P^T: lane (r, g) of the A operand holds V^T[16 d + r][keys 32 st + 8 g .. + 7]
#pragma unroll
        for (int d = 0; d < 6; ++d)                  // rows 16 (d % 3) .. of half d / 3: the halves are contiguous in LDS
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const a16x8 vf = *(const a16x8 *)(ldsV + d * 2048 + vfo[st]);
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    o[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[t][st], o[d][t], 0, 0, 0);
            }
        if constexpr (TRACK) {
            const float wmax = fmaxf(tmax[0], tmax[1]);
            if (__any(wmax > FA64_THR)) {           // wave-uniform, rare
#pragma unroll
                for (int t = 0; t < QT; ++t) (void)move_ref(t, fmaxf(tmax[t], 0.f), true);
            }
        }
    };

    const int ntiles = (Nk + 63) >> 6;
    const bool ragged = (Nk & 63) != 0;
    auto pass = [&](auto track_tag) {
        issue(0, smem);
        if (ntiles > 1) issue(1, smem + FA80_STAGE);
        int slot = 0;
        for (int t = 0; t < ntiles; ++t) {
            // tile t has landed for this wave once at most the DMAs of tile t + 1 are outstanding
            if (t + 1 < ntiles) {
                if (nvb == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();           // every wave's part of tile t is in LDS; tile t - 1 is fully read
            if (t + 2 < ntiles) issue(t + 2, smem + (slot >= 1 ? slot - 1 : 2) * FA80_STAGE);
            const char *cur = smem + slot * FA80_STAGE;
            if (t == 0) first_reference(cur, ragged && ntiles == 1);
            if (ragged && t == ntiles - 1) tile_math(cur, t * 64, std::true_type{}, track_tag);
            else tile_math(cur, t * 64, std::false_type{}, track_tag);
            slot = slot == 2 ? 0 : slot + 1;
        }
    };
    const int dt = d_real >> 4, dq = (d_real >> 2) & 3, de = d_real & 3;
    auto denominator = [&](int t) -> float {         // the riding row sum (half 0): channel d_real of the query's output
        float cand = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (d == dt && e == de) cand = o[d][t][e];
        return __shfl(cand, dq * 16 + r);
    };
    // fast pass / fallback exactly as in flash_attn64_kernel: no tile maxima while every row sum stays inside (2^-100, 2^100)
#ifdef EXP_FA64_ALWAYS_TRACK
    pass(std::true_type{});
    if (false)
#else
    pass(std::false_type{});
#endif
    {
        bool bad = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const uint32_t lb = __builtin_bit_cast(uint32_t, denominator(t));
            bad |= !(lb > 0x0d800000u && lb < 0x71800000u);
        }
        if (__syncthreads_or(bad)) {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                m_run[t] = 0.f;
                if (g == 1) {
                    union { a16x8 v; uint32_t u[4]; } f;
                    f.v = qf[t][1];
                    f.u[0] &= 0xffff0000u;
                    qf[t][1] = f.v;
                }
#pragma unroll
                for (int d = 0; d < 6; ++d) o[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            pass(std::true_type{});
        }
    }

    // ---- normalise and store: lane (g, r) owns channels 16 d + 4 g + (0..3) of query q0 + 16 t + r
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float l = denominator(t);
        const float inv = 1.f / l;
        const int qi = q0 + t * 16 + r;
        if (qi >= Nq) continue;
        a16_t *op = O + ((int64_t)b * Nq + qi) * (H * HV) + h * HV + g * 4;
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = o[d][t][e] * inv;
                if ((d % 3) * 16 + g * 4 + e >= d_real) v[e] = 0.f;
            }
            // padded: rows 16 d of [half 0: 48 | half 1: 48]; COMPACT: only the 40 real dims of each half exist
            if (!COMPACT) *(uint2 *)(op + d * 16) = make_uint2(pack_a2(v[0], v[1]), pack_a2(v[2], v[3]));
            else if ((d % 3) * 16 + g * 4 < 40)
                *(uint2 *)(op + (d / 3) * 40 + (d % 3) * 16) = make_uint2(pack_a2(v[0], v[1]), pack_a2(v[2], v[3]));
        }
    }
}

}  // namespace

// 1 when the kernel onedc_flash_attn picks for (DQ, DV, Nk) reads V^T as bf16 in EITHER build of the library (the d = 40
// kernel: flash_attn64_kernel), 0 when V^T is in the library's own 16-bit format
extern "C" int onedc_flash_attn_vt_is_bf16(int DQ, int DV, int Nk) {
    return ((DQ == 64 && DV == 48) || (DQ == 128 && DV == 96) || (DQ == 40 && DV == 40) || (DQ == 80 && DV == 80)) && Nk >= 64;
}

// q_rs / k_rs: elements between consecutive query / key rows (>= H * DQ: q and k may be column slices of one wider
// projection output, e.g. a fused [q | k] GEMM); vt_bs: elements between the V^T blocks of consecutive images (>= H * DV *
// Nk_pad: the V^T of several layers that share one context can come out of ONE batched GEMM).  0 = dense.
extern "C" int onedc_flash_attn_strided(const void *q, const void *k, const void *vt, void *o, int B, int H, int Nq,
                                        int Nk, int Nk_pad, int DQ, int DV, float scale, int q_rs, int k_rs,
                                        int64_t vt_bs, void *stream) {
    if (!q || !k || !vt || !o || B < 1 || H < 1 || Nq < 1 || Nk < 1) return -1;
    if (Nk_pad < ((Nk + 63) / 64) * 64 || (Nk_pad & 7)) return -2;
    if (q_rs == 0) q_rs = H * DQ;
    if (k_rs == 0) k_rs = H * DQ;
    if (vt_bs == 0) vt_bs = (int64_t)H * DV * Nk_pad;
    if (q_rs < H * DQ || k_rs < H * DQ || vt_bs < (int64_t)H * DV * Nk_pad || (q_rs & 7) || (k_rs & 7) || (vt_bs & 7)) return -4;
    dim3 grid((Nq + 127) / 128, H, B), block(256);
    hipStream_t st = as_stream(stream);
    const float c = scale * 1.4426950408889634f;
#define FA_LAUNCH(dq, dv, ones, pre, dreal)                                                                   \
    hipLaunchKernelGGL((flash_attn_kernel<dq, dv, ones, pre>), grid, block, 0, st, (const a16_t *)q,          \
                       (const a16_t *)k, (const a16_t *)vt, (a16_t *)o, H, Nq, Nk, Nk_pad, c, dreal, q_rs, k_rs, vt_bs)
    // (DQ, DV) = (40, 40) / (80, 80): COMPACT heads (round 5) -- q / k / o rows hold H x d real dims, V^T H x d rows; the padding
    // is made on the DMA path.  Only the two ring kernels read that layout (64 keys and more): -3 otherwise.
    if ((DQ == 40 && DV == 40) || (DQ == 80 && DV == 80)) {
        if (Nk < 64) return -3;
        static OnedcLdsAttrOnce once;
        once.ensure({{(const void *)flash_attn64_kernel<true>, FA64_NST * FA64_STAGE},
                     {(const void *)flash_attn80_kernel<true>, FA64_NST * FA80_STAGE}});
        if (DQ == 40)
            hipLaunchKernelGGL(flash_attn64_kernel<true>, dim3(((Nq + 64 * FA64_QT - 1) / (64 * FA64_QT)) * H * B), block,
                               FA64_NST * FA64_STAGE, st, (const a16_t *)q, (const a16_t *)k, (const a16_t *)vt, (a16_t *)o, H, Nq, Nk,
                               Nk_pad, c, 40, q_rs, k_rs, vt_bs);
        else
            hipLaunchKernelGGL(flash_attn80_kernel<true>, dim3(((Nq + 255) / 256) * H * B), dim3(512), FA64_NST * FA80_STAGE, st,
                               (const a16_t *)q, (const a16_t *)k, (const a16_t *)vt, (a16_t *)o, H, Nq, Nk, Nk_pad, c, 40, q_rs, k_rs,
                               vt_bs);
    } else if (DQ == 64 && DV == 48 && Nk >= 64) {
        static OnedcLdsAttrOnce once;
        once.ensure({{(const void *)flash_attn64_kernel<false>, FA64_NST * FA64_STAGE}});
        hipLaunchKernelGGL(flash_attn64_kernel<false>, dim3(((Nq + 64 * FA64_QT - 1) / (64 * FA64_QT)) * H * B), block, FA64_NST * FA64_STAGE, st,
                           (const a16_t *)q, (const a16_t *)k, (const a16_t *)vt, (a16_t *)o, H, Nq, Nk, Nk_pad, c,
                           40, q_rs, k_rs, vt_bs);
    } else if (DQ == 128 && DV == 96 && Nk >= 64) {
        static OnedcLdsAttrOnce once;
        once.ensure({{(const void *)flash_attn80_kernel<false>, FA64_NST * FA80_STAGE}});
        hipLaunchKernelGGL(flash_attn80_kernel<false>, dim3(((Nq + 255) / 256) * H * B), dim3(512), FA64_NST * FA80_STAGE, st,
                           (const a16_t *)q, (const a16_t *)k, (const a16_t *)vt, (a16_t *)o, H, Nq, Nk, Nk_pad, c,
                           40, q_rs, k_rs, vt_bs);
    } else if (DQ == 64 && DV == 48) FA_LAUNCH(64, 48, true, true, 40);
    else if (DQ == 128 && DV == 96) FA_LAUNCH(128, 96, false, true, 80);      // (fewer than 64 keys; pad dims are zeros)
    else if (DQ == 160 && DV == 160) FA_LAUNCH(160, 160, false, false, 160);
    else if (DQ == 32 && DV == 16) FA_LAUNCH(32, 16, true, true, 8);
    else return -3;
#undef FA_LAUNCH
    ONEDC_LAUNCH_CHECK();
    return 0;
}

extern "C" int onedc_flash_attn(const void *q, const void *k, const void *vt, void *o, int B, int H, int Nq,
                                int Nk, int Nk_pad, int DQ, int DV, float scale, void *stream) {
    return onedc_flash_attn_strided(q, k, vt, o, B, H, Nq, Nk, Nk_pad, DQ, DV, scale, 0, 0, 0, stream);
}
