/* onedc_hip.h — C ABI of the gfx950 (MI355X / CDNA4) kernel library libonedc_hip.so.
 *
 * The reference has no GPU kernels of its own: its device work is stock PyTorch ops issued by
 * the modules on the decode path.  Each entry point below names the reference computation it
 * replaces (paths relative to the reference root, `src/` omitted).  The reference-side binding a
 * maintainer would add is a ctypes stub.
 *
 * Conventions
 *   - raw device pointers, sizes, and a hipStream_t passed as void*; no torch types
 *   - activations are NHWC ("pixel-major"): element (b, y, x, c) of a tensor with channel stride
 *     `cs` and channel offset `co` lives at ((b*H + y)*W + x)*cs + co + c.  A (cs, co) pair lets a
 *     kernel read or write a channel slice of a wider buffer, which is how torch.cat along
 *     channels is realised without copies
 *   - bf16 storage unless a field says f32; all accumulation is f32
 *   - no allocation and no synchronisation inside any entry point (graph-capturable)
 *   - results are deterministic and batch-invariant: no atomics, fixed reduction order (the one split-K form is chosen
 *     by layer shape, never by batch size, and adds its parts in a fixed order)
 *   - return value: 0 ok, <0 argument error, >0 hipError_t of the launch
 */
#ifndef ONEDC_HIP_H
#define ONEDC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* epilogue selector of onedc_conv_gemm: v = act(alpha*acc + bias) [+ residual] */
enum {
    ONEDC_EPI_LINEAR = 0,
    ONEDC_EPI_LEAKY = 1,     /* LeakyReLU(slope)                         dcvc.py:195-197, 246-249 */
    ONEDC_EPI_SILU = 2,
    ONEDC_EPI_GLU_LEAKY = 3, /* LeakyReLU(a,.1)+LeakyReLU(b,.01)         ConvFFN3, dcvc.py:353-367 */
    ONEDC_EPI_GEGLU = 4      /* a * gelu(b)                              diffusers GEGLU (UNet FF) */
};

/* Implicit-GEMM convolution / linear layer on the matrix cores (v_mfma_f32_16x16x32_bf16).
 *   y[m, n] = act(alpha * sum_k x[m, k] * w[n, k] + bias[n]) + res[m, n]
 * m runs over output pixels (b, oy, ox), n over output channels, k over (ky, kx, cin).
 * Replaces every nn.Conv2d / nn.Linear on the decode path: DepthConvBlock4 1x1s (dcvc.py:242-265,
 * 353-367), ResidualBlockUpsample (dcvc.py:183-208), ResnetBlock convs (vqgan/blocks.py:28-33),
 * UNet ResnetBlock2D / Transformer2D projections / up-down samplers (decoder_unet.py:204-299),
 * VAE decoder convs (model_sd15_with_codec_stage1.py:184-188).
 * Weights are pre-packed by the host: bf16 [n_rows_padded][KH*KW*Cin], k-contiguous, rows padded
 * with zeros to a multiple of 128.  For GLU epilogues rows alternate in blocks of 16 (a-block,
 * b-block); for pixel_shuffle rows are ordered (dy*2+dx)*C' + c. */
typedef struct onedc_conv_params {
    const void *x;       /* bf16 input */
    const void *w;       /* bf16 packed weights */
    const float *bias;   /* f32 [n] or NULL (GLU: [2][n/2] laid out like the weight rows) */
    const void *res;     /* bf16 residual on the output pixel grid, or NULL */
    void *y;             /* bf16 (or f32 when out_f32) output */
    const void *zeros;   /* >= 256 B of zeros in device memory (source of halo / tail rows) */
    float *stats;        /* NULL, or GroupNorm partial sums of the stored output for the consumer's norm:
                            float2 (sum, sum of squares) per (64-pixel block, stats_gran channels) of the y buffer,
                            [B*OH*OW/64][y_cs/stats_gran]; needs OH*OW % 64 == 0, y_co % 4 == 0, no pixel_shuffle.
                            Replaces the statistics pass of nn.GroupNorm (see onedc_gn_from_stats) */
    int32_t B, IH, IW, Cin, x_cs, x_co; /* Cin: channels read, multiple of 64 */
    int32_t OH, OW, Cout, y_cs, y_co;   /* Cout: GEMM columns (before GLU halving / shuffle) */
    int32_t r_cs, r_co;
    int32_t KH, KW, stride, pad, up;    /* up=1: nearest x2 upsample of the input fused in (any kernel);
                                           up=2: the same layer (3x3, stride 1, pad 1) as four 2x2 phase convolutions on
                                           the low-resolution input: w is [4 Cout rows][4 taps x Cin] from
                                           pack.upsample_phase_matrix, 4/9 of the multiply-adds (gemm_conv_ups.hip;
                                           needs Cout % 128 == 0, IH, IW % 16 == 0, plain bf16 epilogue, no residual) */
    int32_t epi, out_f32, pixel_shuffle;
    float alpha, slope;
    int32_t batch;                      /* >1: batched GEMM over blockIdx.z */
    int32_t tile;                       /* 0: pick the kernel automatically.  Forcing one (tests, benchmarks):
                                           128 = 4-wave kernels (128x128 / 64x128 / 16x256 tiles),
                                           256 = 8-wave 256x256 tile, 129 = 8-wave 128(ch)x256(px) tile,
                                           130 = 3x3 window kernel (16x16-pixel patches, stride 1, pad 1), 8 waves,
                                           131 = the same layer on the 4-wave window kernel (two workgroups per CU),
                                           132 = as 128 with the epilogue's scattered 8-byte stores instead of line stores,
                                           133 = 4-wave tiles on conv_lin_kernel whatever K (1x1 layers), four-stage ring,
                                           134 = the same with the three-stage ring on the 128 x 128 tile (three workgroups
                                                 per CU: the automatic choice for K <= 384) */
    int32_t stats_gran;                 /* channels per statistics entry: 4 (quads), or 2 (pairs) for buffers whose
                                           GroupNorm groups are not multiples of 4 channels (320, 960 channels) */
    int32_t patch16;                    /* set by onedc_conv_gemm (callers leave 0): pixel tiles are 16x16 patches */
    int32_t in_f32;                     /* 1: x, w, res are f32 (exact f32 MFMA; Cin multiple of 32): the prior
                                           networks, whose outputs feed integer decisions, and the VAE's conv_in;
                                           2 (version 0.4): x and res f32 as with 1, w in the split form of
                                           pack.pack_matrix_split16 -- per row and 32-channel block 32 halves h then 32 halves
                                           l with w = h + l * 2^-11, channels in the kernel's slot order -- and the products
                                           on the 16-bit matrix cores (three v_mfma_f32_16x16x32_f16 per block: 22-bit
                                           operands, exact products, f32 accumulation; |x| < 65504) */
    int64_t x_bs, w_bs, y_bs, r_bs;     /* element strides between batches */
    const float *gn_ab;                 /* NULL, or float2 (a, b) [B][Cin] from onedc_gn_from_stats / onedc_gn_stats: x is the
                                           RAW tensor in front of nn.GroupNorm (+SiLU when gn_silu) and the conv reads
                                           silu(a*x + b) -- the normalisation is applied while the operand is staged
                                           (only the 3x3 window kernel does it: ask onedc_conv_win_takes first) */
    int32_t gn_silu;
    int32_t out_bf16;                   /* 1: a 16-bit output is stored as bf16 whatever the library's activation format (the
                                           IEEE-half build hands V^T to onedc_flash_attn's d = 40 kernel this way: its P / V^T
                                           product stays bf16 for the exponent range); ignored with out_f32 */
    /* ---- version 0.4 (fields appended: a caller that zero-fills the struct keeps the 0.3 behaviour) ---- */
    void *ws;                           /* NULL, or device scratch of ws_bytes for the split-K form below (the caller owns
                                           it; one per stream) */
    int64_t ws_bytes;
    int32_t split_k;                    /* 0: the dispatcher decides BY LAYER SHAPE ONLY (never by B: a picture decodes to
                                           the same bits alone and inside a batch); 1: never split; n > 1: n parts where the
                                           layer qualifies.  Split form: the K-tiles of a 3x3 layer on a map of <= 256
                                           pixels (the UNet's 12x12 level: 2 304 pixels per batch of 16 against 11 520 or
                                           23 040 k) are cut into n contiguous ranges, one workgroup per (tile, range)
                                           leaves f32 partial sums in ws, onedc_splitk_reduce adds them in range order
                                           (fixed order, no atomics) and applies the epilogue.  Ask onedc_conv_splitk_ws_bytes
                                           for the scratch size (0: this layer is not split).  ABI 5: a layer that qualifies
                                           and finds ws NULL or smaller than that size is REFUSED with -15 -- it is never run
                                           unsplit behind the caller's back (the unsplit kernel adds in another order, so a
                                           fixed-size scratch would make a picture's bits depend on the batch it sits in);
                                           a caller without scratch sets split_k = 1 */
    const void *x2;                     /* NULL, or the input of a 1x1 convolution that is accumulated into the same output
                                           tile behind the 3x3 taps: y = alpha * (conv3x3(x) + conv1x1(x2)) + bias + res --
                                           diffusers ResnetBlock2D.conv_shortcut riding on conv2 (called from
                                           model_sd15_with_codec_stage1.py:184-188).  Same pixel grid as y, 16-bit NHWC; only the
                                           4-wave window kernel takes it: ask onedc_conv_win_takes_x2 first */
    const void *w2;                     /* packed [rows padded to 128][Cin2] weights of that 1x1 layer, divided by alpha by
                                           the host where alpha != 1 (a power of two there) */
    int32_t Cin2, x2_cs, x2_co;         /* Cin2: multiple of 32 */
    /* ---- round 5 (appended: a caller that zero-fills the struct keeps the 0.4 behaviour) ---- */
    int32_t x_bf16;                     /* 1 (IEEE-half build; ignored by the bf16 build): the 3x3 taps multiply bf16 OPERANDS -- x is a
                                           normalised tensor stored as bf16 (onedc_scale_shift_act_fmt with out_bf16) or, with gn_ab,
                                           is normalised to bf16 on load; w is packed bf16 -- while y, res, stats, x2 and w2 keep the
                                           build's format.  The chip holds a higher clock on 8-bit mantissas; an intermediate that only
                                           this conv reads costs no stored precision.  Only the 4-wave window kernel does it: any other
                                           layer is refused with -17 (ask onedc_conv_gemm_kernel_id first) */
} onedc_conv_params;
/* bytes of scratch (onedc_conv_params.ws) the split-K form of this layer needs; 0: the dispatcher does not split it */
int64_t onedc_conv_splitk_ws_bytes(const onedc_conv_params *p);
/* 1 if onedc_conv_gemm would run p (x2 / w2 / Cin2 / x2_cs / x2_co filled in) on the window kernel with the second source */
int onedc_conv_win_takes_x2(const onedc_conv_params *p);
int onedc_conv_gemm(const onedc_conv_params *p, void *stream);
/* 1 if onedc_conv_gemm would run p on the 3x3 window kernel AND that kernel can apply a GroupNorm in front of the conv
 * while it stages the operand (gn_ab; p->gn_ab itself may still be NULL when asking): same conditions as the launch */
int onedc_conv_win_takes(const onedc_conv_params *p);
/* The kernel family onedc_conv_gemm runs p on (the dispatcher's own decision, without launching), or the negative
 * status it would return.  bench.py labels its per-kernel timings with this. */
enum {
    ONEDC_KERNEL_WIN3X3 = 1,       /* conv_win_kernel: 16x16-pixel patches x 128 channels, window staged once per chunk */
    ONEDC_KERNEL_UPS2X2 = 2,       /* conv_ups_kernel: x2 upsample + 3x3 as four 2x2 phase convolutions */
    ONEDC_KERNEL_256X256 = 3,      /* conv_gemm256_kernel */
    ONEDC_KERNEL_128X256 = 4,      /* conv_gemm128x256_kernel */
    ONEDC_KERNEL_16X256 = 5,       /* conv_gemm_kernel<Tile<16,256>> */
    ONEDC_KERNEL_64X128 = 6,       /* conv_gemm_kernel<Tile<64,128>> */
    ONEDC_KERNEL_128X128 = 7,      /* conv_gemm_kernel<Tile<128,128>> */
    ONEDC_KERNEL_F32_128X128 = 8,  /* conv_gemm_kernel<Tile<128,128>, f32> (exact-f32 path) */
    ONEDC_KERNEL_F32_64X128 = 9,   /* conv_gemm_kernel<Tile<64,128>, f32>: the same for grids of less than two rounds */
    ONEDC_KERNEL_WIN3X3_4W = 10,   /* conv_win4_kernel: the window kernel as 4-wave workgroups, two per CU */
    ONEDC_KERNEL_UPS2X2_4W = 11,   /* conv_win4_kernel<2>: the upsample phases on the same 4-wave machinery */
    ONEDC_KERNEL_SPLITK_256 = 12,  /* conv_gemm256_kernel over K ranges + splitk_reduce_kernel (onedc_conv_params.split_k) */
    ONEDC_KERNEL_LIN_64X128 = 13,  /* conv_lin_kernel<Tile<64,128>>: 1x1 layers, 32-channel K-steps, four-stage LDS ring */
    ONEDC_KERNEL_LIN_128X128 = 14, /* conv_lin_kernel<Tile<128,128>> */
    ONEDC_KERNEL_F32S_128X128 = 15,/* conv_gemm_kernel<Tile<128,128>, f32, split>: f32 activations, operands as pairs of halves */
    ONEDC_KERNEL_F32S_64X128 = 16,
    ONEDC_KERNEL_LIN_128X256 = 17  /* conv_lin_kernel<Tile<128,256,1,4>, 3 stages>: 128 channels x 256 pixels on four waves (wave tile
                                      128 x 64), 72 KB: two workgroups per CU; 25 % fewer L2 -> LDS bytes per flop than 128 x 128 */
};
int onedc_conv_gemm_kernel_id(const onedc_conv_params *p);
const char *onedc_conv_kernel_name(int kernel_id);

/* Small-channel direct convolution (Cin < 64: VAE conv_in / post_quant_conv, HyperDecoder.feat_in
 * codec_module.py:149-152).  w: f32 [Cout][KH][KW][Cin]; x bf16 or f32 (x_f32). */
typedef struct onedc_direct_conv_params {
    const void *x; const float *w; const float *bias; void *y;
    int32_t B, H, W, Cin, x_cs, x_co, Cout, y_cs, y_co, K, x_f32, out_f32, epi;
    float slope, in_scale;              /* x is multiplied by in_scale first (1/0.18215 for the VAE) */
} onedc_direct_conv_params;
int onedc_direct_conv(const onedc_direct_conv_params *p, void *stream);

/* 3x3 / stride 1 / pad 1 convolution with at most 4 output channels, optionally with the GroupNorm (+SiLU) in front of
 * it applied while the input is staged: AutoencoderKL decoder.conv_norm_out + SiLU + conv_out (128 -> 3 at picture
 * resolution; src/models/sd15_onedc_codec_stage1/model_sd15_with_codec_stage1.py:184-188).  x bf16 NHWC view; w the
 * layer's packed bf16 matrix [rows >= Cout][9*Cin] (same packing as onedc_conv_gemm); gn_ab NULL or float2 [B][Cin]
 * from onedc_gn_stats / onedc_gn_from_stats; y f32 (y_f32) or bf16.  H, W % 16 == 0, Cin % 32 == 0 (-1 otherwise);
 * H * W * x_cs < 2^31 (-2: the caller runs the scale/shift pass and onedc_conv_gemm instead). */
int onedc_conv3x3_few_out(const void *x, const void *w, const float *bias, const float *gn_ab, void *y, int B, int H,
                          int W, int Cin, int x_cs, int x_co, int Cout, int y_cs, int y_co, int y_f32, int gn_silu,
                          void *stream);

/* Depthwise 3x3, pad 1 (DepthConv.depth_conv, dcvc.py:251).  w f32 [3][3][C], bias f32 [C]. */
int onedc_dwconv3x3(const void *x, const float *w, const float *bias, void *y, int B, int H, int W,
                    int C, int x_cs, int x_co, int y_cs, int y_co, void *stream);
/* the same on f32 activations (prior networks) */
int onedc_dwconv3x3_f32(const float *x, const float *w, const float *bias, float *y, int B, int H, int W,
                        int C, int x_cs, int x_co, int y_cs, int y_co, void *stream);

/* GroupNorm(32 groups) [+ SiLU] as statistics + a per-channel affine:
 *   onedc_gn_stats        per-chunk partial sums (fixed order) -> per image/group mean, rstd ->
 *                         ab[b][c] = (rstd*gamma, beta - mean*rstd*gamma), f32 [B][C][2]
 *   onedc_scale_shift_act y = act(x * a + bb), act = SiLU or identity
 * Replaces nn.GroupNorm + SiLU in ResnetBlock (vqgan/blocks.py:39-46), reduce_resblock
 * (decoder_unet.py:19-26), diffusers ResnetBlock2D / Transformer2D.norm / Attention.group_norm.
 * "Image" is whatever the caller makes it: for the VAE mid-block attention, whose statistics are
 * per 16x16 window (vae/autoencoders_patch_attn.py:22-28), the windows are the batch.
 * partial: f32 workspace of B * onedc_gn_chunks(HW) * 64 floats (not touched for maps of at most 256 pixels whose groups are
 * whole 8-channel units: those are reduced in one launch). */
int onedc_gn_chunks(int HW);
int onedc_gn_stats(const void *x, const float *gamma, const float *beta, float *partial, float *ab, int B,
                   int HW, int C, int x_cs, int x_co, float eps, void *stream);
int onedc_scale_shift_act(const void *x, const float *ab, void *y, int B, int HW, int C, int x_cs, int x_co,
                          int y_cs, int y_co, int silu, void *stream);
/* the same; out_bf16 = 1 stores y as bf16 whatever the library's format (the consumer is a conv with onedc_conv_params.x_bf16) */
int onedc_scale_shift_act_fmt(const void *x, const float *ab, void *y, int B, int HW, int C, int x_cs, int x_co,
                              int y_cs, int y_co, int silu, int out_bf16, void *stream);
/* the statistics step from partial sums left by the producing conv's epilogue (onedc_conv_params.stats):
 * stats [B*HW/64][cs/gran] float2 for the whole buffer (cs channels per pixel), normalised channels
 * [co, co + C); C/32 and co multiples of gran (4 or 2).  Writes ab like onedc_gn_stats. */
int onedc_gn_from_stats(const float *stats, const float *gamma, const float *beta, float *ab, int B, int HW,
                        int C, int cs, int co, float eps, int gran, void *stream);

/* LayerNorm over the last dim of [rows][C] bf16 (BasicTransformerBlock norm1-3). */
int onedc_layernorm(const void *x, const float *gamma, const float *beta, void *y, int64_t rows,
                    int C, float eps, void *stream);

/* Row softmax of f32 scores [rows][n] -> bf16 probabilities [rows][n_pad] (AttnBlock,
 * vqgan/blocks.py:95; VAE mid attention).  Columns n..n_pad-1 are written as 0. */
int onedc_softmax_rows(const float *s, void *p, int64_t rows, int n, int n_pad, void *stream);

/* Flash attention for the UNet's multi-head self/cross attention (diffusers Attention with
 * SDPA; call sites decoder_unet.py:204-293).  q [B][Nq][H*DQ], k [B][Nk][H*DQ] bf16,
 * vt [B][H*DV][Nk_pad] bf16 (V transposed, keys contiguous), o [B][Nq][H*DV].
 * (DQ, DV) in {(64,48), (128,96), (160,160), (32,16)}: head dims 40 / 80 / 160 (decoder UNet) and 8 (the
 * analysis UNet2DModel of the encode side), zero-padded by the packed projection weights.  Head dim 80 is laid out as two
 * half-heads of the d = 40 layout: q / k dims [0..39 | 24 zeros | 40..79 | 24 zeros], v rows [0..39 | 8 zeros | 40..79 |
 * 8 zeros] (onedc_amd/pack.py: head_positions); o has the v layout.  scale multiplies q.k before the softmax.
 * COMPACT heads (round 5): (DQ, DV) = (40, 40) or (80, 80) with Nk >= 64 -- q / k / o rows hold H x d REAL dims back to back, vt
 * H x d rows; the kernels build the padded LDS images themselves (padding chunks and rows are DMA'd from a page of zeros, like
 * the constant-1 chunk and the ones row of the softmax trick), so the projections in front and behind compute and move no padded
 * columns.  Results are bit-identical to the padded layout.  Fewer than 64 keys: -3 (use the padded layout). */
int onedc_flash_attn(const void *q, const void *k, const void *vt, void *o, int B, int H, int Nq,
                     int Nk, int Nk_pad, int DQ, int DV, float scale, void *stream);
/* The same with strides: q_rs / k_rs = elements between consecutive query / key rows (q, k as column slices of one wider
 * projection output: a fused [to_q | to_k] GEMM, or the to_k of all layers that share a context), vt_bs = elements between
 * the V^T blocks of consecutive images (the to_v of several layers from one batched GEMM); 0 = dense.  Multiples of 8. */
int onedc_flash_attn_strided(const void *q, const void *k, const void *vt, void *o, int B, int H, int Nq, int Nk,
                             int Nk_pad, int DQ, int DV, float scale, int q_rs, int k_rs, int64_t vt_bs, void *stream);
/* 1 if onedc_flash_attn reads vt as bf16 for this shape whatever the library's activation format (head dims 40 and 80 with
 * at least 64 keys: P and V^T of those kernels are bf16, see attention.hip); the producer sets onedc_conv_params.out_bf16 */
int onedc_flash_attn_vt_is_bf16(int DQ, int DV, int Nk);

/* The cross-attention half of a diffusers BasicTransformerBlock as ONE launch (round 6; reference call site
 * src/models/sd15_onedc_codec_stage1/decoder_unet.py:204-293, where every transformer block of the UNet attends to the same 144
 * context tokens):      y = h + to_out( softmax( to_q(LayerNorm(h)) . K^T * scale ) . V ) + to_out.bias
 * replacing onedc_layernorm + onedc_conv_gemm (to_q) + onedc_flash_attn_strided + onedc_conv_gemm (to_out, residual) and three
 * passes of the residual stream through HBM: a workgroup keeps 64 pixels resident from the LayerNorm to the residual add
 * (csrc/xattn_block.hip).  h, y: [B * HW][*_cs] 16-bit rows, channels [*_co, *_co + C) (y must not alias h: a tile's rows are
 * read again for the residual late in the workgroup).  ln_g / ln_b / bo: f32 [C].  wq, wo: the layers' packed matrices [C rounded up to 128 rows][C], k-contiguous, 16-bit
 * (onedc_amd/pack.py: pack_matrix; to_q has no bias).  k: [B * Nk][k_rs] 16-bit, head i at columns i * d (COMPACT heads);
 * vt: [B][heads * d][Nk_pad] with vt_bs elements between pictures, zero beyond Nk, bf16 when vt_bf16 (what
 * onedc_flash_attn_vt_is_bf16 says the flash kernel of this shape reads: the projection GEMM stores it once for both) else the
 * library's format.  Shapes: C = 320, heads = 8, d = 40, Nk % 16 == 0 and <= 160, HW % 64 == 0, strides multiples of 8;
 * onedc_xattn_block_ok answers 1 / 0 for a parameter set without launching, onedc_xattn_block returns -1 for a refused one. */
typedef struct onedc_xattn_params {
    const void *h; const float *ln_g; const float *ln_b; const void *wq; const void *k; const void *vt; const void *wo;
    const float *bo; void *y;
    int32_t B, HW, C, heads, d, Nk, Nk_pad, k_rs;
    int32_t h_cs, h_co, y_cs, y_co, vt_bf16;
    float ln_eps, scale;
    int64_t vt_bs;
    const float *bq;                    /* NULL, or a bias of the q projection, f32 [C].  A host may fold the LayerNorm's affine into to_q --
                                           wq = Wq . diag(gamma), bq = Wq . beta -- and pass ln_g = ln_b = NULL: the kernel then only centres and
                                           scales the tile (40 fewer loads and 80 fewer multiply-adds per thread in front of the first MFMA) */
} onedc_xattn_params;
int onedc_xattn_block_ok(const onedc_xattn_params *p);
int onedc_xattn_block(const onedc_xattn_params *p, void *stream);

/* ---- elementwise heads ------------------------------------------------------------------- */
/* FSQ.indices_to_codes, levels [4]*7 (codec_module.py:431): idx int32 [n] -> f32 codes [n][8]
 * (7 used, channel 7 = 0). */
int onedc_fsq_codes(const int32_t *idx, float *codes, int64_t n, void *stream);
/* FSQ.forward quantisation of the hyper latent on the encode side (codec_module.py:379; same levels):
 * z NHWC (bf16 / f32) with channel stride z_cs, the 7 dims at offset z_co -> int32 indices [n] */
int onedc_fsq_quantize(const void *z, int z_f32, int z_cs, int z_co, int32_t *idx, int64_t n, void *stream);

/* One step k of the four-part prior (compression_model.py:369-407 decode, :303-358 encode).
 * params: bf16/f32 NHWC [B][H][W][2C] = (scales | means), C = 128.
 *  - prior_indexes: scale plane of step k -> int16 rANS table indexes [B][C/4][H][W] (NCHW order
 *    of the reference's flattening; combine_for_writing :296-301 + build_indexes
 *    entropy_models.py:355-362)
 *  - prior_update_dec: y_hat += (symbols + means) * mask_k        (:383-385)
 *  - prior_update_enc: symbols = round((y - means)*mask_k) combined; y_hat += that + means*mask_k
 *  - prior_update_zonly: y_hat += means * mask_k                  (:410-418)
 * y_hat is f32 NHWC [B][H][W][C] inside a buffer with channel stride y_cs. */
int onedc_prior_indexes(const void *params, int params_f32, int16_t *indexes, int B, int H, int W,
                        int C, int step, void *stream);
int onedc_prior_update_dec(const void *params, int params_f32, const int16_t *symbols, void *y_hat,
                           int yhat_f32, int y_cs, int B, int H, int W, int C, int step, void *stream);
int onedc_prior_update_enc(const void *params, int params_f32, const float *y, int16_t *symbols,
                           void *y_hat, int yhat_f32, int y_cs, int B, int H, int W, int C, int step,
                           void *stream);
int onedc_prior_update_zonly(const void *params, int params_f32, void *y_hat, int yhat_f32, int y_cs,
                             int B, int H, int W, int C, int step, void *stream);

/* get_x0_from_noise (dmd/utils.py:279-284) in f64 like the reference call site
 * (model_sd15_with_codec_stage1.py:320-322): x0 = (reduced - sqrt(1-a)*eps)/sqrt(a). f32 in/out. */
int onedc_x0_from_noise(const float *reduced, const float *eps, float *x0, int64_t n,
                        double alpha_bar, void *stream);

/* NHWC (bf16/f32) padded picture -> NCHW f32 [B][3][H][W] crop (model...stage1.py:327-329),
 * optional clamp(-1,1)*0.5+0.5 (inference.py:106).
 * nonfinite: NULL, or int32 [B] zero-filled by the caller; word b is set to 1 when picture b holds a NaN / Inf BEFORE the
 * clamp (which would turn it into a valid-looking pixel): the range guard of the IEEE-half build. */
int onedc_image_out_checked(const void *x, int x_f32, int x_cs, float *out, int B, int PH, int PW, int H, int W,
                            int unit_range, int32_t *nonfinite, void *stream);
/* the same without the range flag: the signature of library versions up to 0.2, kept so that a host built against it
 * keeps working (version 0.3 had inserted `nonfinite` in front of `stream` under the old name) */
int onedc_image_out(const void *x, int x_f32, int x_cs, float *out, int B, int PH, int PW, int H, int W, int unit_range,
                    void *stream);
/* The same crop as 8-bit samples, converted on the device: clamp(-1,1)*0.5+0.5 (inference.py:106), then x255 + 0.5,
 * clamp(0,255), truncation (torchvision save_image, inference.py:107,120).  nhwc = 1: [B][H][W][3] (what an image
 * writer takes), 0: [B][3][H][W].  A batch then crosses PCIe as 1 byte per sample. */
int onedc_image_out_u8(const void *x, int x_f32, int x_cs, uint8_t *out, int B, int PH, int PW, int H, int W,
                       int nhwc, int32_t *nonfinite, void *stream);

/* generic strided copy / cast between NHWC views (channel concat, padding layouts) */
int onedc_copy_channels(const void *x, int x_f32, int x_cs, int x_co, void *y, int y_f32, int y_cs,
                        int y_co, int64_t pixels, int C, void *stream);
/* window partition for the VAE mid attention (vae/autoencoders_patch_attn.py:22-28): the region
 * [y0,y0+RH) x [x0,x0+RW) of a (B,H,W,C) map <-> (B*nwin, wh*ww, C) windows.  The reference's slicing
 * leaves ragged windows at the right/bottom edge; the caller covers them as extra regions. */
int onedc_window_partition(const void *x, void *y, int B, int H, int W, int C, int y0, int x0, int RH,
                           int RW, int wh, int ww, int reverse, void *stream);
/* ---- encode side ------------------------------------------------------------------ */
/* NCHW f32 picture -> NHWC view (f32 / bf16), padded on the right / bottom from (H, W) to (Hp, Wp): replicate
 * (model_sd15_with_codec_stage1.py:281-285) or, reflect = 1, reflect (z-only harness,
 * models/sd15_onedc_codec_z_only/inference.py:106-109); channels beyond C of the destination are not written. */
int onedc_nchw_to_nhwc(const float *x, void *y, int y_f32, int B, int C, int H, int W, int Hp, int Wp, int y_cs,
                       int y_co, int reflect, void *stream);
/* input of Encoder.pix_emb (8x8 / stride 8 conv, codec_module.py:60) as a 1x1 GEMM operand:
 * NHWC f32 picture (3 channels, stride x_cs) -> bf16 [B][H/8][W/8][192], k = (ky, kx, c) */
int onedc_patchify8(const float *x, int x_cs, void *y, int B, int H, int W, void *stream);
/* latent_dist.sample() * scaling_factor: moments f32 [n][8], noise f32 [n][4] or NULL (mode) -> f32 [n][4] */
int onedc_latent_sample(const float *moments, const float *noise, float *out, int64_t n, float scale, void *stream);

/* ---- the prior networks in the reference's CPU arithmetic (compatibility path; csrc/ref_arith.hip) ------------------------
 * torch-CPU's f32 convolutions are fixed arrangements of fused-multiply-add chains (measured: DESIGN.md section 5); these entry
 * points evaluate the same chains with v_fma_f32, so that a prior built from them takes the table-index decisions of the
 * reference that wrote a stream (compression_model.py:369-407, entropy_models.py:355-362).  All tensors f32, NHWC views.
 *   onedc_ref_conv1x1   y = act(conv1x1(x)) [+ res]: wT [C][OC] (transposed weights); input channels in blocks of `blk`: the
 *                       first block one chain that starts from the bias (bias_first) or from 0, later blocks chains from 0
 *                       added in order; when !bias_first the bias is added to the first block's sum (native gemm path).  act 0 none, 1 LeakyReLU(slope), 2 ConvFFN3's gate
 *                       LeakyReLU(rows n, slope) + LeakyReLU(rows n + OC/2, slope2) (dcvc.py:353-367: OC/2 outputs);
 *                       shuffle = 1 stores through F.pixel_shuffle(., 2) (dcvc.py:183-208; y is the (2H, 2W) map, no res)
 *   onedc_ref_conv3x3   dense 3x3, pad 1, w3T [9][C][OC]: (icb 16, one_chain 0) oneDNN avx512: bias + per 16 input channels a chain
 *                       from 0 over (ky, kx, channel); (icb 8, one_chain 1) oneDNN avx2: one chain from the bias over (8-channel
 *                       block, ky, kx, channel); y = [LeakyReLU(slope)](.) [+ res]
 *   onedc_ref_conv3x3_im2col  the same layer where torch takes its native path (maps of <= 20 480 elements: im2col + sgemm):
 *                       wKT [9 C][OC] with k = c * 9 + ky * 3 + kx; K in `parts` equal parts, blocks of `blk` inside a part
 *                       (chains from 0 added in order), bias joined to part 0, parts [0, split) and [split, parts) summed in
 *                       order and then added (split == parts: one sequence)
 *   onedc_ref_prior_indexes  build_indexes with the writer's logarithm: index = number of thresholds <= sigma
 *                       (thresholds: f32 [n], ascending, found by the host by bisection over torch.log) */
int onedc_ref_conv1x1(const float *x, int x_cs, int x_co, const float *wT, const float *bias, const float *res, int r_cs,
                      int r_co, float *y, int y_cs, int y_co, int64_t pixels, int C, int OC, int blk, int bias_first,
                      int act, float slope, float slope2, int shuffle, int H, int W, void *stream);
int onedc_ref_conv3x3(const float *x, int x_cs, int x_co, const float *w3T, const float *bias, const float *res, int r_cs,
                      int r_co, float *y, int y_cs, int y_co, int B, int H, int W, int C, int OC, int leaky_on, float slope,
                      int icb, int one_chain, void *stream);
int onedc_ref_conv3x3_im2col(const float *x, int x_cs, int x_co, const float *wKT, const float *bias, const float *res, int r_cs,
                             int r_co, float *y, int y_cs, int y_co, int B, int H, int W, int C, int OC, int parts, int blk,
                             int split, int leaky_on, float slope, void *stream);
int onedc_ref_prior_indexes(const float *params, const float *thresholds, int n_thresholds, int16_t *indexes, int B, int H,
                            int W, int C, int step, void *stream);

/* MXFP8 experiment (reference call the layers belong to: model_sd15_with_codec_stage1.py:184-188): quantise every
 * block of 32 consecutive channels of every pixel to OCP e4m3 with one shared power-of-two scale (OCP MX v1.0: what
 * v_mfma_scale_f32_16x16x128_f8f6f4 consumes along K) and write the values back dequantised, 16 bit -> 16 bit.  A conv that
 * reads y with weights treated the same way computes what an MXFP8 layer would (exact products, f32 accumulation), so the
 * picture-level cost of such a layer is measured on hardware before any kernel is written for it.  C % 32 == 0. */
int onedc_mxfp8_roundtrip(const void *x, void *y, int64_t pixels, int C, int x_cs, int x_co, int y_cs, int y_co,
                          void *stream);

/* sinusoidal timestep embedding, flip_sin_to_cos, shift 0 (decoder_unet.py:138): f32 [n][dim] */
int onedc_timestep_embedding(const float *t, float *emb, int n, int dim, void *stream);
/* y = a + b (bf16), used where a residual cannot ride on a GEMM epilogue */
int onedc_add(const void *a, const void *b, void *y, int64_t n, void *stream);

const char *onedc_hip_version(void);
/* ABI number of this header: bumped whenever a signature, a struct field's meaning or an entry point's behaviour changes in a
 * way an older caller could not notice at link time (symbol-compatible breaks).  History: 2 = library 0.2; 3 = 0.3
 * (onedc_image_out gained `nonfinite` in front of `stream` UNDER THE OLD NAME); 4 = 0.4 (onedc_image_out back to its 0.2
 * signature, the 0.3 form renamed onedc_image_out_checked; fields appended to onedc_conv_params); 5 = round 5 (a split-K
 * layer without scratch is refused with -15 instead of run unsplit; onedc_hip_abi_version itself).  A host checks
 * onedc_hip_abi_version() == ONEDC_HIP_ABI_VERSION once after dlopen: a host compiled against ABI 3 that calls
 * onedc_image_out of a 4+ library would pass its flag pointer where the stream is read. */
#define ONEDC_HIP_ABI_VERSION 5
int onedc_hip_abi_version(void);
/* identity (sha256 prefix) of the kernel sources, headers and compiler flags this library was built from; the Python loader
 * refuses a library whose id differs from the sources beside it (set ONEDC_ALLOW_STALE_LIB=1 for experiment builds) */
const char *onedc_hip_build_id(void);
/* "bf16" or "f16": the 16-bit activation format this build of the library computes in (exact token) */
const char *onedc_hip_act_format(void);

#ifdef __cplusplus
}
#endif
#endif /* ONEDC_HIP_H */
