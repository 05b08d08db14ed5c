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
__global__ __launch_bounds__(256) void flash_attn_kernel(const bf16_t *__restrict__ Q,
                                                         const bf16_t *__restrict__ K,
                                                         const bf16_t *__restrict__ VT,
                                                         bf16_t *__restrict__ O, int H, int Nq, int Nk,
                                                         int Nk_pad, float scale_log2e, int d_real) {
    using C = FA<DQ, DV, ONES_ROW>;
    constexpr int NBUF = PREFETCH ? 2 : 1;
    __shared__ __attribute__((aligned(16))) char lds[NBUF * C::STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;

    // Q fragments (B operand): lane (g, r) holds Q[q0 + 16*t + r][32*ks + 8*g .. +7]
    bf16x8 qf[2][C::QS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = q0 + t * 16 + r;
#pragma unroll
        for (int ks = 0; ks < C::QS; ++ks) {
            if (qi < Nq)
                qf[t][ks] = *(const bf16x8 *)(Q + ((int64_t)b * Nq + qi) * (H * DQ) + h * DQ + ks * 32 + g * 8);
            else
                qf[t][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    f32x4 o[C::DT][2];
#pragma unroll
    for (int d = 0; d < C::DT; ++d)
#pragma unroll
        for (int t = 0; t < 2; ++t) o[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

    const bf16_t *Kb = K + (int64_t)b * Nk * (H * DQ) + h * DQ;
    const bf16_t *Vb = VT + ((int64_t)b * H + h) * DV * Nk_pad;
    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);

    uint4 kreg[C::NKL], vreg[C::NVL];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < C::NKL; ++i) {
            const int c = tid + 256 * i;
            const int row = c / C::KCH, ch = c - row * C::KCH;
            kreg[i] = make_uint4(0, 0, 0, 0);
            if ((C::KCHUNKS % 256 == 0 || c < C::KCHUNKS) && k0 + row < Nk)
                kreg[i] = *(const uint4 *)(Kb + (int64_t)(k0 + row) * (H * DQ) + ch * 8);
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
                const bf16x8 kf = *(const bf16x8 *)(ldsK + (kt * 16 + r) * C::KROW + (ks * 4 + g) * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    s[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ks], s[kt][t], 0, 0, 0);
            }
        }
        // ---- online softmax, lane-local per query column
        bf16x8 pf[2][2];   // [t][step]: P^T B-operand fragments
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
                union { bf16x8 v; uint32_t u[4]; } pk;
                pk.u[0] = pack_bf2(s[2 * st][t][0], s[2 * st][t][1]);
                pk.u[1] = pack_bf2(s[2 * st][t][2], s[2 * st][t][3]);
                pk.u[2] = pack_bf2(s[2 * st + 1][t][0], s[2 * st + 1][t][1]);
                pk.u[3] = pack_bf2(s[2 * st + 1][t][2], s[2 * st + 1][t][3]);
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
                union { bf16x8 v; uint2 h2[2]; } vf;
                vf.h2[0] = *(const uint2 *)vp;
                vf.h2[1] = *(const uint2 *)(vp + 32);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    o[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[t][st], o[d][t], 0, 0, 0);
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
        bf16_t *op = O + ((int64_t)b * Nq + qi) * (H * DV) + h * DV + g * 4;
#pragma unroll
        for (int d = 0; d < C::DT; ++d) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = o[d][t][e] * inv;
                if (ONES_ROW && d * 16 + g * 4 + e >= d_real) v[e] = 0.f;   // keep the pad channels zero
            }
            *(uint2 *)(op + d * 16) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        }
    }
}

}  // namespace

extern "C" int onedc_flash_attn(const void *q, const void *k, const void *vt, void *o, int B, int H, int Nq,
                                int Nk, int Nk_pad, int DQ, int DV, float scale, void *stream) {
    if (!q || !k || !vt || !o || B < 1 || H < 1 || Nq < 1 || Nk < 1) return -1;
    if (Nk_pad < ((Nk + 63) / 64) * 64 || (Nk_pad & 7)) return -2;
    dim3 grid((Nq + 127) / 128, H, B), block(256);
    hipStream_t st = as_stream(stream);
    const float c = scale * 1.4426950408889634f;
#define FA_LAUNCH(dq, dv, ones, pre, dreal)                                                                   \
    hipLaunchKernelGGL((flash_attn_kernel<dq, dv, ones, pre>), grid, block, 0, st, (const bf16_t *)q,          \
                       (const bf16_t *)k, (const bf16_t *)vt, (bf16_t *)o, H, Nq, Nk, Nk_pad, c, dreal)
    if (DQ == 64 && DV == 48) FA_LAUNCH(64, 48, true, true, 40);
    else if (DQ == 96 && DV == 80) FA_LAUNCH(96, 80, false, true, 80);
    else if (DQ == 160 && DV == 160) FA_LAUNCH(160, 160, false, false, 160);
    else return -3;
#undef FA_LAUNCH
    ONEDC_LAUNCH_CHECK();
    return 0;
}
