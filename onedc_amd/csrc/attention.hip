// attention.hip — flash attention for the UNet's multi-head self / cross attention.
//
// One workgroup = 4 wave64s = 128 queries of one (batch, head); each wave owns 32 queries.
// Everything is computed TRANSPOSED so the softmax is lane-local:
//   S^T = K . Q^T     (MFMA A = K tile rows from LDS, B = Q fragments held in registers)
//   O^T += V^T . P^T  (MFMA A = V^T tile rows from LDS, B = P^T built from the S^T accumulators
//                      in registers: with the query on the lane, the accumulator of one product is
//                      already the B operand of the next, in a permuted but consistent key order)
// A lane owns one query column: its running max / sum / rescale never leave the lane except for a
// two-step butterfly over the four lane quads that split the keys.
// K/V tiles of 64 keys are staged through registers into padded LDS rows (the guide's
// recommendation for attention; LDS-DMA cannot pad).  V arrives transposed ([channel][key]) from
// the projection GEMM, which produces it in that layout for free by swapping its operands.
#include "common.h"

namespace {

template <int DQ, int DV>
__global__ __launch_bounds__(256) void flash_attn_kernel(const bf16_t *__restrict__ Q,
                                                         const bf16_t *__restrict__ K,
                                                         const bf16_t *__restrict__ VT,
                                                         bf16_t *__restrict__ O, int H, int Nq, int Nk,
                                                         int Nk_pad, float scale) {
    constexpr int KROW = DQ * 2 + 16;      // padded LDS row bytes of the K tile
    constexpr int VROW = 64 * 2 + 16;      // padded LDS row bytes of the V^T tile (64 keys)
    constexpr int QS = DQ / 32;            // k-steps of the S product
    constexpr int DT = DV / 16;            // output tiles along d
    constexpr int KCH = DQ / 8;            // 16-byte chunks per K row
    __shared__ __attribute__((aligned(16))) char lds[64 * KROW + DV * VROW];
    char *ldsK = lds;
    char *ldsV = lds + 64 * KROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;

    // Q fragments (B operand): lane (g, r) holds Q[q0 + 16*t + r][32*ks + 8*g .. +7]
    bf16x8 qf[2][QS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = q0 + t * 16 + r;
#pragma unroll
        for (int ks = 0; ks < QS; ++ks) {
            if (qi < Nq)
                qf[t][ks] = *(const bf16x8 *)(Q + ((int64_t)b * Nq + qi) * (H * DQ) + h * DQ + ks * 32 + g * 8);
            else
                qf[t][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    f32x4 o[DT][2];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int t = 0; t < 2; ++t) o[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

    const bf16_t *Kb = K + (int64_t)b * Nk * (H * DQ) + h * DQ;
    const bf16_t *Vb = VT + ((int64_t)b * H + h) * DV * Nk_pad;

    for (int k0 = 0; k0 < Nk; k0 += 64) {
        __syncthreads();
        // ---- stage K tile [64][DQ] and V^T tile [DV][64]
        for (int c = tid; c < 64 * KCH; c += 256) {
            const int row = c / KCH, ch = c - row * KCH;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (k0 + row < Nk) v = *(const uint4 *)(Kb + (int64_t)(k0 + row) * (H * DQ) + ch * 8);
            *(uint4 *)(ldsK + row * KROW + ch * 16) = v;
        }
        for (int c = tid; c < DV * 8; c += 256) {
            const int row = c >> 3, ch = c & 7;
            *(uint4 *)(ldsV + row * VROW + ch * 16) =
                *(const uint4 *)(Vb + (int64_t)row * Nk_pad + k0 + ch * 8);
        }
        __syncthreads();

        // ---- S^T tiles: s[kt][t] = keys 16*kt..+15 (rows) x queries of tile t (cols)
        f32x4 s[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int t = 0; t < 2; ++t) s[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < QS; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const bf16x8 kf = *(const bf16x8 *)(ldsK + (kt * 16 + r) * KROW + (ks * 4 + g) * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    s[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ks], s[kt][t], 0, 0, 0);
            }
        }

        // ---- online softmax, lane-local per query column
        bf16x8 pf[2][2];   // [t][step]: P^T B-operand fragments
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float mx = -1e30f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = k0 + kt * 16 + g * 4 + e;
                    float v = s[kt][t][e] * scale;
                    if (key >= Nk) v = -1e30f;
                    s[kt][t][e] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[t], mx);
            const float alpha = __expf(m_run[t] - m_new);
            m_run[t] = m_new;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __expf(s[kt][t][e] - m_new);
                    s[kt][t][e] = pv;
                    sum += pv;
                }
            l_run[t] = l_run[t] * alpha + sum;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[d][t][e] *= alpha;
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

        // ---- O^T += V^T . P^T ; A fragment of lane (r, g): V^T[16*d + r][keys 32*st + 4g + (0..3)
        //      and 32*st + 16 + 4g + (0..3)] -- the same permuted key order as pf
#pragma unroll
        for (int d = 0; d < DT; ++d) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const char *vp = ldsV + (d * 16 + r) * VROW + (st * 32 + g * 4) * 2;
                union { bf16x8 v; uint2 h2[2]; } vf;
                vf.h2[0] = *(const uint2 *)vp;
                vf.h2[1] = *(const uint2 *)(vp + 32);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    o[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[t][st], o[d][t], 0, 0, 0);
            }
        }
    }

    // ---- normalise and store: lane (g, r) owns channels 16*d + 4g + (0..3) of query q0 + 16*t + r
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float l = l_run[t];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;
        const int qi = q0 + t * 16 + r;
        if (qi >= Nq) continue;
        bf16_t *op = O + ((int64_t)b * Nq + qi) * (H * DV) + h * DV + g * 4;
#pragma unroll
        for (int d = 0; d < DT; ++d)
            *(uint2 *)(op + d * 16) =
                make_uint2(pack_bf2(o[d][t][0] * inv, o[d][t][1] * inv), pack_bf2(o[d][t][2] * inv, o[d][t][3] * inv));
    }
}

}  // namespace

extern "C" int onedc_flash_attn(const void *q, const void *k, const void *vt, void *o, int B, int H, int Nq,
                                int Nk, int Nk_pad, int DQ, int DV, float scale, void *stream) {
    if (!q || !k || !vt || !o || B < 1 || H < 1 || Nq < 1 || Nk < 1) return -1;
    if (Nk_pad < ((Nk + 63) / 64) * 64 || (Nk_pad & 7)) return -2;
    dim3 grid((Nq + 127) / 128, H, B), block(256);
    hipStream_t st = as_stream(stream);
#define FA(dq, dv)                                                                                      \
    hipLaunchKernelGGL((flash_attn_kernel<dq, dv>), grid, block, 0, st, (const bf16_t *)q, (const bf16_t *)k, \
                       (const bf16_t *)vt, (bf16_t *)o, H, Nq, Nk, Nk_pad, scale)
    if (DQ == 64 && DV == 48) FA(64, 48);
    else if (DQ == 96 && DV == 80) FA(96, 80);
    else if (DQ == 160 && DV == 160) FA(160, 160);
    else return -3;
#undef FA
    ONEDC_LAUNCH_CHECK();
    return 0;
}
