/* tdr.h -- C ABI of libtdr_hip.so: the MI355X (gfx950) kernels behind the
 * guided-restoration train step of TextualDegRemoval (NAFNet-ref path).
 *
 * The reference has no FFI for this path (it is pure PyTorch); each entry point
 * below replaces the ATen call sites of one SURVEY.md section-8a row and cites
 * them (paths relative to the reference checkout).  Conventions:
 *   - caller allocates every buffer (device pointers, fp32 contiguous NCHW,
 *     int32/int64 where stated); the library never owns or frees memory;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*) and are
 *     asynchronous; no global mutable state;
 *   - return 0 on success, negative error code otherwise, never throws;
 *     tdr_last_error() returns a thread-local message.
 *   - "ns" arguments are per-image strides in floats (so channel slices /
 *     concat halves of a bigger buffer can be addressed without copies).
 */
#ifndef TDR_H
#define TDR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* C-ABI version: bumped with every incompatible change of this header (100 rounds 1-2, 101 round 3, 102 round 4, 103 - 104
 * round 5, 105 - 107 round 6: the SFNet operators, their inference modes, the table-driven finishing reductions); the
 * binding (textualdegremoval_amd/_lib.py) refuses a library whose version differs from the one it was written against. */
#define TDR_ABI_VERSION 107
int tdr_version(void);
const char* tdr_last_error(void);

/* Device-resident verdict of a train step (10 words).  The fp16-split backward pass runs on gradients scaled by the
 * power of two `scale` (tdr_l1_loss_guarded emits dpred * scale, tdr_multi_copy_guarded multiplies the parameter
 * gradients by inv_scale -- both exact).  tdr_grad_sumsq_guarded sets `finite` from the global gradient norm: a
 * non-finite norm (an operand left the fp16 range, or the forward pass overflowed) makes tdr_adamw_step_guarded a
 * no-op for that step, halves `scale` and counts the step in `skipped`; after `growth_interval` finite steps the scale
 * doubles again up to max_scale.  growth_interval < 0 switches the verdict OFF: every step is applied (finite = 1) and the scale never
 * moves -- the modes without a loss scale (TDR_MATH=bx3 / f32: the reference's arithmetic) use the struct only as the device-resident
 * step counter.  `step` is AdamW's t (applied steps only) and bc1/bc2_sqrt its bias corrections.
 * Everything lives in device memory so the whole step replays from captured hipGraphs with step-invariant launch
 * arguments; the reference has no counterpart (it runs torch fp32, image_restoration_ref_model.py:276-279). */
typedef struct TdrStepGuard {
    float scale, inv_scale, max_scale;
    int good, growth_interval;
    int step, skipped, finite;
    float bc1, bc2_sqrt;
} TdrStepGuard;

/* ---------------------------------------------------------------------------
 * Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32,
 * exact fp32).  One kernel family serves every dense conv on the path:
 *   1x1 (NAFBlock conv1/3/4/5, sca)      network_nafnet_guided_arch.py:183-205
 *   3x3 s1/s2 (+ReLU)  Encoder/ResidualBlock/intro/ending   :44-59,116-129,429-434
 *   2x2 s2 (downs) / 1x1+PixelShuffle (ups)                 :449-451,468-473
 *   dilated 3x3 correlation of `search`, 3x3 correlation of `search_org` :495-536
 * and their data-gradients (same kernel, weights packed transposed/flipped).
 * out[n,m,oy,ox] = epi( sum_{c,ky,kx} Wp[m,(c,ky,kx)] * B[n,c,oy*s+ky*d-pad,ox*s+kx*d-pad] )
 *   B = in            (gate=0)
 *   B = in[c]*in[c+Cin]   (gate=1: SimpleGate fused into the operand load, :170-175)
 *   B *= kscale[n*kscale_ns + c]       (SCA channel scale / beta / gamma folded in)
 * epi STD    : v=acc; +bias[n*bias_ns+m]; *scale[n*scale_ns+m]; +bias2_mul*bias2[n*bias2_ns+m];
 *              +res[n,m,oy,ox]; relu; zero where mask[n,m,oy,ox]<=0
 * epi GATEBWD: out[m]=acc*aux[m+Cout], out[m+Cout]=acc*aux[m]   (SimpleGate backward)
 * epi PSHUF  : out[m/4, 2oy+(m%4)/2, 2ox+m%2] = acc (+res there)  (PixelShuffle(2) / stride-2 dgrad)
 * ------------------------------------------------------------------------- */
typedef struct TdrConvDesc {
    int N, Cin, H, W;            /* input tensor (Cin = K-channels actually contracted) */
    int Cout, OH, OW;            /* GEMM rows and output grid (before PSHUF) */
    int KH, stride, dil, pad;    /* square kernel, pad = low-side padding */
    const float* in;  int64_t in_ns;
    int gate;
    const float* kscale; int64_t kscale_ns;
    const void* wp;  int64_t wp_ns;  int Mpad;    /* packed weights, see tdr_pack_weights / tdr_pack_weights_bx3 */
    int wp_fmt;                  /* 0: fp32 rows (tdr_pack_weights, exact fp32 MFMA)
                                    1: 3-way bf16 split fragments (tdr_pack_weights_bx3, bf16 MFMA x6, fp32-equivalent)
                                    2: 2-way fp16 split fragments (tdr_pack_weights_hx2, f16 MFMA x3, operands in fp16 range)
                                    3: the same pack read as plain fp16 (head plane only, ONE f16 MFMA product, fp32 accumulate):
                                       reduced precision, BASELINE configs[4]'s "fp16 MFMA" arithmetic (TDR_MATH=h1) */
    float* out; int64_t out_ns;
    int epi;                     /* 0 STD, 1 GATEBWD, 2 PSHUF */
    const float* bias;  int64_t bias_ns;
    const float* scale; int64_t scale_ns;
    const float* bias2; int64_t bias2_ns; float bias2_mul;
    const float* res;   int64_t res_ns;
    const float* mask;  int64_t mask_ns;
    const float* aux;   int64_t aux_ns;
    int relu;                    /* 0 none, 1 ReLU, 2 exact (erf) GELU, 3 quick_gelu x*sigmoid(1.702x) */
} TdrConvDesc;

int tdr_conv_forward(const TdrConvDesc* d, void* stream);

/* Packed weight layout consumed by tdr_conv_forward:
 *   Wp[chunk][tap][ck][Mpad],  chunk = c / CK, ck = c % CK, CK = tdr_conv_ck(KH_eff)
 * mode 0 FWD      : M=Cout, c=ci,  Wp = W[m][c][tap]
 * mode 1 DGRAD_S1 : M=Cin,  c=co,  Wp = W[c][m][taps-1-tap]        (stride-1 data gradient)
 * mode 2 DGRAD_2x2S2 : M=4*Cin (m=ci*4+a*2+b), c=co, 1x1: Wp = W[c][ci][a][b]  (use with PSHUF)
 * mode 3 DGRAD_3x3S2 : M=4*Cin, c=co, 2x2 taps (u,v): W[c][ci][ky(a,u)][kx(b,v)] or 0 (use with PSHUF)
 * w is the standard contiguous (Cout,Cin,KH,KH) tensor. */
int tdr_conv_ck(int KH_eff);
int64_t tdr_packed_weight_floats(int M, int Kch, int KH_eff);
int tdr_pack_weights(const float* w, int Cout, int Cin, int KH, int mode, float* wp, void* stream);
/* Split-bf16 packing for the bf16 matrix-core path (wp_fmt = 1): every weight is split into three bf16
 * terms h+m+l (24+ significant bits) and laid out in MFMA A-fragment order
 *   Wp3[c/16][tap][m/32][split][lane][8]  (16-byte fragments; lane = (m%32) + 32*((c%16)/8), element = c%8)
 * same `mode` semantics as tdr_pack_weights.  The convolution then evaluates each fp32 product as six
 * bf16 cross products with fp32 accumulation (v_mfma_f32_32x32x16_bf16), see csrc/tdr_conv_bx3.hip. */
int64_t tdr_packed_weight_bytes_bx3(int M, int Kch, int KH_eff);
int tdr_pack_weights_bx3(const float* w, int Cout, int Cin, int KH, int mode, void* wp, void* stream);
/* the same for B matrices in one launch (per-image weights, TdrConvDesc.wp_ns = packed bytes / 4): matrix b is read at
 * w + b*w_stride floats and packed to wp + b*tdr_packed_weight_bytes_bx3(M, Kch, KH) bytes; mode 0 or 1 */
int tdr_pack_weights_bx3_batch(const float* w, int64_t w_stride, int B, int Cout, int Cin, int KH, int mode, void* wp,
                               void* stream);
/* 2-way fp16 split (wp_fmt 2): [group][tap][m-tile][split h,m][lane] 16-byte fragments of 8 x f16; three f16 MFMA products
 * (hh, hm, mh) per fp32 product.  fp32-class accuracy for operands inside the fp16 range -- forward activations and
 * weights; gradient-sized operands need an exact power-of-two pre-scale (profiles/r1/fp16x2_probe_mi355x.log). */
int64_t tdr_packed_weight_bytes_hx2(int M, int Kch, int KH_eff);
int tdr_pack_weights_hx2(const float* w, int Cout, int Cin, int KH, int mode, void* wp, void* stream);
/* tuning aid (profiles/autotune_conv.py): force tile configuration `cfg` (0 = built-in heuristic) of the split-bf16
 * forward kernels with kernel size kh == 1, or of the 3x3 / 2x2 ones (any other kh) */
int tdr_conv_force_cfg(int kh, int cfg);
/* Multi-tensor packing: one launch packs every (weight, mode, format) job of a step.
 * The caller fills jobs with tdr_pack_job_init (host), sets first_block = running sum of ceil(total/256),
 * copies the array to the device and launches with total_blocks = the final sum. */
typedef struct TdrPackJob {
    const float* w; void* wp;
    int Cout, Cin, KH, mode, fmt;      /* fmt 0: fp32 rows, 1: split-bf16 fragments */
    int M, Kch, KHe, CK, Mx;           /* derived by tdr_pack_job_init (Mx = Mpad for fmt 0, m-tiles for fmt 1) */
    int64_t total;                     /* work items (floats for fmt 0, 16-byte fragment triples for fmt 1) */
    int64_t first_block;
} TdrPackJob;
int tdr_pack_job_init(TdrPackJob* job, const float* w, int Cout, int Cin, int KH, int mode, int fmt, void* wp);
int tdr_pack_weights_multi(const TdrPackJob* jobs_dev, int n_jobs, int64_t total_blocks, void* stream);
/* Per-image "filters" of search / search_org: 3x3 patches cut from LR blocks become GEMM rows.
 * blk [B][G][C][BH][BW]; M = G*PH*PW rows, m = g*PH*PW + py*PW + px;
 * Wp[b][c][tap] = blk[b,g,c, py*pstep+ky*dil+off, px*pstep+kx*dil+off]  (3x3 taps, CK=8 layout,
 * per-image stride = tdr_packed_weight_floats(M, C, 3)). */
int tdr_pack_patches(const float* blk, int B, int G, int C, int BH, int BW, int PH, int PW, int pstep, int dil,
                     int off, float* wp, void* stream);

/* ---------------------------------------------------------------------------
 * Pre-split activations ("P16" tensors) and the 3x3 convolution that consumes them (csrc/tdr_conv_p16.hip).
 * Replaces the 3x3 / stride 1 / pad 1 convolutions of the MASA encoder's ResidualBlocks, forward and data gradient
 * (network_nafnet_guided_arch.py:44-59,110-143), in the 2-way fp16 split arithmetic.
 * P16 image of an fp32 [N][C][H][W] tensor, C % 16 == 0: 16-byte slots [N][C/8][plane][H+2][W+2], a slot = 8 consecutive
 * channels of one pixel as 8 x f16; plane 0 = rn_f16(x) (an exact zero is stored as -0.0: the sign bit of plane 0 is "x <= 0" even
 * where a tiny positive x rounds to +0), plane 1 = rn_f16(x - rn_f16(x)); the 1-pixel border is zero.
 * tdr_p16_bytes: buffer size.  tdr_p16_from_f32 / tdr_p16_to_f32 convert (to_f32 returns head + residual).
 * Plane formats (ABI 104).  fmt 2: two fp16 planes (above; TDR_MATH=hx2, operands inside the fp16 window).  fmt 1: THREE bf16 planes
 * h = rn_bf16(x), m = rn_bf16(x - h), l = rn_bf16(x - h - m), 6 bytes per element: 8 + 8 + 8 significand bits on fp32's exponent, so
 * h + m + l == x exactly for every normal fp32 x -- the tensor IS the fp32 tensor (no window, no loss scale; TDR_MATH=bx3: the
 * reference's arithmetic, models/image_restoration_ref_model.py:268-279), products lh hl mm mh hm hh on v_mfma_f32_32x32x16_bf16.
 * tdr_conv3x3_p16 / tdr_wgrad3x3_p16 take the format from wp_fmt / fmt; every plane tensor of a call has that format.
 * tdr_conv3x3_p16: out = mask( relu( conv(in, W) + bias + res ) ); `in` is a P16 tensor of Cin channels, `wp` an hx2 pack
 * (tdr_pack_weights_hx2, mode 0 forward / mode 1 data gradient; wp_fmt must be 2); the residual and the ReLU mask (> 0) are read
 * from fp32 NCHW tensors (res32 / mask32, strides in floats) or from P16 tensors of Cout channels (res16 / mask16: the mask is
 * the sign bit of the head plane, see above); the result is written as fp32 NCHW (out32), as a P16 tensor including its zero border (out16),
 * or both.  Accumulation order = tdr_conv_forward's 2-way fp16 split kernel: bit-identical results on the same operands. */
typedef struct TdrConvP16Desc {
    int N, Cin, H, W, Cout;
    const void* in;
    const void* wp; int Mpad; int wp_fmt;
    const float* bias;
    const float* res32;  int64_t res32_ns;
    const void*  res16;
    const float* mask32; int64_t mask32_ns;
    const void*  mask16;
    int relu;
    float* out32; int64_t out32_ns;
    void*  out16;
} TdrConvP16Desc;
int64_t tdr_p16_bytes(int N, int C, int H, int W);
int tdr_p16_from_f32(const float* src, int64_t src_ns, int N, int C, int H, int W, void* dst, void* stream);
int tdr_p16_to_f32(const void* src, int N, int C, int H, int W, float* dst, int64_t dst_ns, void* stream);
/* the same for either plane format (fmt 1: bf16 triple, 2: fp16 pair; the three above are fmt 2) */
int64_t tdr_p16_bytes_fmt(int N, int C, int H, int W, int fmt);
int tdr_p16_from_f32_fmt(const float* src, int64_t src_ns, int N, int C, int H, int W, void* dst, int fmt, void* stream);
int tdr_p16_to_f32_fmt(const void* src, int N, int C, int H, int W, float* dst, int64_t dst_ns, int fmt, void* stream);
int tdr_conv3x3_p16(const TdrConvP16Desc* d, void* stream);
/* tuning aid: force a tile configuration (0 = heuristic) */
int tdr_conv3x3_p16_force_cfg(int cfg);
/* Weight (and bias) gradient of the same convolution from P16 operands (csrc/tdr_wgrad_p16.hip):
 *   g[co][ci][ky][kx] = sum_{n,y,x} dout[n,co,y,x] * in[n,ci,y+ky-1,x+kx-1],   db[co] = sum dout[n,co,y,x]  (optional)
 * in16: P16 tensor of Cin channels (the convolution's input), dout16: P16 tensor of Cout channels (the output gradient, inside the
 * fp16 window: loss-scaled backward).  Deterministic: split-K partials in ws are reduced in a fixed order. */
typedef struct TdrWgradP16Desc {
    int N, Cin, H, W, Cout;
    const void* in16;
    const void* dout16;
    float* g;
    float* db;
    float* ws; int64_t ws_floats;
    int fmt;                     /* plane format of in16 and dout16: 2 fp16 pair (0 is read as 2), 1 bf16 triple */
} TdrWgradP16Desc;
int64_t tdr_wgrad3x3_p16_ws_floats(const TdrWgradP16Desc* d);
int tdr_wgrad3x3_p16(const TdrWgradP16Desc* d, void* stream);

/* ---------------------------------------------------------------------------
 * Weight gradient GEMM (K = pixels) on the fp32 matrix cores.
 *   G[g][co][ci][tap] = sum_{n in group g} sum_{oy,ox} dout[n,co,oy,ox] * B[n,ci,oy*s+ky-pad,ox*s+kx-pad]
 * groups = 1 (sum over the batch) or N (per-image, needed by the SCA/beta chain).
 * Deterministic: split-K partials in `ws` are reduced in fixed order.
 * ------------------------------------------------------------------------- */
typedef struct TdrWgradDesc {
    int N, Cin, H, W, Cout, OH, OW, KH, stride, pad;
    const float* in;   int64_t in_ns;  int gate;
    const float* dout; int64_t dout_ns;
    float* g;          /* [groups][Cout][Cin][KH*KH] */
    float* db;         /* optional [Cout]: bias gradient sum_{n,oy,ox} dout (fused, saves a pass over dout) */
    int per_image;
    float* ws; int64_t ws_floats;      /* split-K workspace */
    int math;          /* 0: exact fp32 MFMA; 1: 3-way bf16 split on the bf16 MFMA pipe where supported
                          (stride 1, 1x1 / 3x3), exact fp32 otherwise; 2: 2-way fp16 split (3 products) on the same
                          kernels -- both operands must lie in the fp16 range (activations; gradients of a loss-scaled
                          backward pass, see tdr_l1_loss); 3: plain fp16 (one product), same range requirement, reduced precision.
                          With math >= 2 the stride-2 cases (3x3 pad 1, 2x2 pad 0, H = 2 OH, W = 2 OW) run on the split as well
                          (csrc/tdr_wgrad_s2.hip) instead of falling through to the exact kernel */
} TdrWgradDesc;
int64_t tdr_wgrad_ws_floats(const TdrWgradDesc* d);
int tdr_conv_wgrad(const TdrWgradDesc* d, void* stream);

/* Grouped 1x1 weight gradients (round 5): the leaf weight gradients of a whole level of NAFBlocks -- autograd's dW / db of conv1, conv4,
 * conv5 of every block, models/archs/network_nafnet_guided_arch.py:183-205,216-238 -- have the same shape and nothing downstream in
 * the backward pass reads them, so they run as ONE launch + ONE fixed-order reduction after the data-gradient chain (deterministic).
 * `d` carries the common shape (N, Cin, Cout, H = OH, W = OW, KH = 1, in_ns, dout_ns, gate, math 1 or 2); its pointers are ignored.
 * `table`: TdrWg1GroupEntry[nprob] in DEVICE memory; entry p: the operands, the problem's own workspace of
 * tdr_wgrad1x1_group_ws_floats(d, nprob) floats (`part`; `dbpart` = part + bpp * Cout * Cin with bpp = (ws_floats / (Cout * (Cin + 1))),
 * or NULL for no bias gradient) and the outputs g [Cout][Cin], db [Cout] or NULL. */
typedef struct TdrWg1GroupEntry {
    const float* in; const float* dout; float* part; float* dbpart; float* g; float* db;
} TdrWg1GroupEntry;
int tdr_wgrad1x1_group_supported(const TdrWgradDesc* d);
int64_t tdr_wgrad1x1_group_ws_floats(const TdrWgradDesc* d, int nprob);
int tdr_wgrad1x1_group(const TdrWgradDesc* d, int nprob, const void* table, void* stream);

/* ---------------------------------------------------------------------------
 * Streaming (HBM-bound) kernels.
 * ------------------------------------------------------------------------- */
/* LayerNorm2d fwd/bwd -- models/archs/nafnet_arch_utils.py:264-300 (eps 1e-6) and
 * network_restormer_guided_arch.py:172-218 (eps 1e-5; WithBias = center 1, BiasFree = center 0, b NULL).
 * center 1: y = w*(x-mu)*rstd+b ; center 0: y = w*x*rstd (variance still about the mean).
 * mu,rstd [N,H*W] saved for backward.
 * bwd (center 1): gx = rstd*(g - yhat*mean_c(g*yhat) - mean_c(g)) (+add), g=go*w;
 *      gw = sum go*yhat, gb = sum go  (deterministic two-stage; ws >= tdr_ln_ws_floats)
 * bwd (center 0): gx = rstd*(g - yhat*mean_c(g*x*rstd)) (+add); gw = sum go*x*rstd; gb is scratch. */
int tdr_layernorm2d_fwd(const float* x, int64_t x_ns, const float* w, const float* b, float eps, int center,
                        int N, int C, int HW, float* y, float* mu, float* rstd, void* stream);
int64_t tdr_ln_ws_floats(int N, int C, int HW);
int tdr_layernorm2d_bwd(const float* go, const float* x, int64_t x_ns, const float* mu, const float* rstd,
                        const float* w, const float* add, int64_t add_ns, int add_C, int center,
                        int N, int C, int HW, float* gx, float* gw, float* gb, float* ws, void* stream);

/* depthwise 3x3 (+bias) + SimpleGate + global-average-pool partials
 *   network_nafnet_guided_arch.py:185-187,170-175,192-196
 * t [N,2C,H,W] -> g [N,C,H,W] = dw(t)[:C]*dw(t)[C:],  pooled[N,C] = mean_hw g */
int64_t tdr_dwsg_ws_floats(int N, int C, int H, int W);
int tdr_dwsg_fwd(const float* t, const float* w, const float* b, int N, int C, int H, int W,
                 float* g, float* pooled, float* ws, void* stream);
/* backward: dg [N,C,H,W] -> dt [N,2C,H,W], dw [2C,9], db [2C] */
int tdr_dwsg_bwd(const float* dg, const float* t, const float* w, const float* b, int N, int C, int H, int W,
                 float* dt, float* dw, float* db, float* ws, void* stream);
/* same, the incoming gradient being dg[n][c][:] + dg_bias[n][c] * dg_bias_mul (the SCA branch's pooled gradient / HW,
 * which the reference's autograd adds through the adaptive-average-pool backward, :192-196) */
int tdr_dwsg_bwd_biased(const float* dg, const float* dg_bias, float dg_bias_mul, const float* t, const float* w, const float* b,
                        int N, int C, int H, int W, float* dt, float* dw, float* db, float* ws, void* stream);
/* dw = db = NULL in the two calls above (allowed when tdr_dwsg_bwd_parts_supported(W): the one-pass backward): the per-workgroup
 * partials of the parameter gradients stay at the start of ws and the caller finishes them later -- they are leaves of the backward
 * pass -- with tdr_dw_param_finish on the same (N, C, H, W).  (ABI 103) */
int tdr_dwsg_bwd_parts_supported(int W);
int tdr_dw_param_finish(const float* ws, int N, int C, int H, int W, float* dw /*[2C,9]*/, float* db /*[2C]*/, void* stream);

/* ---- Restormer-ref depthwise stencils (models/archs/network_restormer_guided_arch.py); b / db may be NULL (bias=False).
 * GDFN gate (:236-239): t [N,2C,H,W] -> g [N,C,H,W] = gelu(dw(t)[:C]) * dw(t)[C:]  (erf GELU) */
int tdr_dwgelu_fwd(const float* t, const float* w, const float* b, int N, int C, int H, int W, float* g, void* stream);
/* dg [N,C,H,W] -> dt [N,2C,H,W], dw [2C,9], db [2C]; ws >= tdr_dwsg_ws_floats(N,C,H,W) */
int tdr_dwgelu_bwd(const float* dg, const float* t, const float* w, const float* b, int N, int C, int H, int W,
                   float* dt, float* dw, float* db, float* ws, void* stream);
/* plain depthwise 3x3, pad 1 (MDTA qkv_dwconv :254,260): t [N,planes,H,W] -> out same shape; planes even, W % 4 == 0 */
int tdr_dwconv_fwd(const float* t, const float* w, const float* b, int N, int planes, int H, int W, float* out,
                   void* stream);
/* dout -> dt = dw^T(dout), dw [planes,9], db [planes]; ws >= tdr_dwsg_ws_floats(N,planes/2,H,W) */
int tdr_dwconv_bwd(const float* dout, const float* t, const float* w, int N, int planes, int H, int W, float* dt,
                   float* dw, float* db, float* ws, void* stream);
/* the same pair with a trailing ReLU (DRSformer-ref MSFN, network_drsformer_guided_arch.py:226-253: relu(dwconv3x3(x))):
 * relu != 0 clamps the forward output; act (the saved forward output, may be NULL) masks dout in the backward. */
int tdr_dwconv_act_fwd(const float* t, const float* w, const float* b, int N, int planes, int H, int W, int relu, float* out,
                       void* stream);
int tdr_dwconv_act_bwd(const float* dout, const float* act, const float* t, const float* w, int N, int planes, int H, int W,
                       float* dt, float* dw, float* db, float* ws, void* stream);
/* grouped 3x3 with two inputs per output channel (DRSformer-ref MSFN dwconv3x3_1, network_drsformer_guided_arch.py:231-232,
 * 246-247): t [N][2C][H][W], w [C][2][3][3], b [C] | NULL, out [N][C][H][W] = relu?(dw(t[2c]) + dw(t[2c+1]) + b[c]).
 * Backward in one pass (W <= 1024); act: the saved forward output when relu was set, else NULL; ws: tdr_dwsg_ws_floats.
 * out / dout / act may be channel slices of wider tensors: *_ns is their per-image stride in floats (multiples of 4). */
int tdr_dwpair_fwd(const float* t, const float* w, const float* b, int N, int C, int H, int W, int relu, float* out,
                   int64_t out_ns, void* stream);
int tdr_dwpair_bwd(const float* dout, int64_t dout_ns, const float* act, int64_t act_ns, const float* t, const float* w, int N,
                   int C, int H, int W, float* dt, float* dw, float* db, float* ws, void* stream);
/* the plain depthwise 3x3 (+ ReLU) whose 2C output planes live in two tensors (MSFN's cross-concatenation, :244-247: x1 = [a3[:h] |
 * a5[:h]], x2 = [a3[h:] | a5[h:]] is written in place instead of copied): planes [0, C) at outA, planes [C, 2C) at outB, both
 * [N][C][H][W] with per-image stride out_ns.  The one-pass backward reads dout / act split the same way (actA = actB = NULL: no ReLU). */
int tdr_dwconv_halves_fwd(const float* t, const float* w, const float* b, int N, int planes, int H, int W, int relu, float* outA,
                          float* outB, int64_t out_ns, void* stream);
int tdr_dwconv_halves_bwd(const float* doutA, const float* doutB, int64_t dout_ns, const float* actA, const float* actB,
                          int64_t act_ns, const float* t, const float* w, int N, int planes, int H, int W, float* dt, float* dw,
                          float* db, float* ws, void* stream);

/* ---- Restormer-ref MDTA core (:246-277), per image and head over CHANNEL tokens (c = C/heads <= 192).
 * The pixel contractions run on tdr_conv_wgrad (per_image Gram q k^T) and tdr_conv_forward (1x1, per-image weights);
 * these entry points do the c x c part.  Cp = tdr_mdta_pad(C) = C rounded up to 32.
 * out[n][r] = sum_p x[n][r][p]^2  (F.normalize denominators of q and k, :266-267) */
int tdr_row_sumsq(const float* x, int64_t x_ns, int N, int rows, int HW, float* out, void* stream);
int tdr_mdta_pad(int C);
/* G [N,C,C] (G[i][j] = q_i . k_j), ss [N,2C] (|q|^2 then |k|^2), temp [heads] ->
 * A [N,Cp,Cp]: A[i][j] = softmax_j(temp_h * G_ij / (max(|q_i|,1e-12) max(|k_j|,1e-12))) inside a head, 0 elsewhere;
 * AT = A^T.  Both are directly the fp32 packed 1x1 weights ([cin][Mpad]) of tdr_conv_forward:
 * wp = AT computes attn v, wp = A computes attn^T dout. */
int tdr_mdta_softmax(const float* G, const float* ss, const float* temp, int N, int C, int heads, float* A, float* AT,
                     void* stream);
/* dA [N,C,C] (dA[i][j] = dout_i . v_j) -> W [N,Wp,Wp] (Wp = tdr_mdta_pad(2C)), the symmetric packed 1x1 weights with
 * d[q;k] = W [q;k], and dtemp [heads].  ws >= N*heads floats. */
int tdr_mdta_bwd(const float* G, const float* ss, const float* temp, const float* A, const float* dA, int N, int C,
                 int heads, float* W, float* dtemp, float* ws, void* stream);
/* out = a * alpha[0] + b  (b may be NULL): TransformerResFusionBlock `x * alpha + shortcut` (:353) and its backward */
int tdr_axpby_dev(const float* a, const float* alpha, const float* b, int64_t numel, float* out, void* stream);
/* out[0] = sum a*b (fixed-order two-stage, double accumulation of the partials); ws >= 512 floats */
int tdr_dot(const float* a, const float* b, int64_t numel, float* out, float* ws, void* stream);
/* nn.PixelShuffle(2): in [N,4C,H,W] -> out [N,C,2H,2W] (:391; backward of Downsample's PixelUnshuffle :378) */
int tdr_pixel_shuffle2(const float* in, int N, int C, int H, int W, float* out, void* stream);

/* SCA 1x1 on the pooled vector: s[n,co] = sum_ci Wsca[co,ci]*pooled[n,ci] + bsca[co] (:192-196) */
int tdr_sca_fwd(const float* pooled, const float* wsca, const float* bsca, int N, int C, float* s, void* stream);
/* Per-block parameter-gradient epilogue of the conv3/SCA/beta chain.  Inputs:
 * G3 [N,C,C] per-image sum_pix dy[co]*g[ci], S3[C]=sum dy.  Outputs dW3,db3,dbeta,
 * dWsca,dbsca and dpooled[N,C] (to be added /HW to dg). */
int tdr_sca_bwd(const float* G3, const float* S3, const float* w3, const float* b3, const float* beta,
                const float* s, const float* pooled, const float* wsca, int N, int C,
                float* dw3, float* db3, float* dbeta, float* dwsca, float* dbsca, float* dpooled,
                float* ws /* N*C floats */, void* stream);
/* conv5/gamma chain: G5 [Cout,C] = sum dout*g2 (unscaled), S5[Cout]:
 * dW5=gamma*G5, db5=gamma*S5, dgamma=sum_ci W5*G5 + b5*S5 */
int tdr_scaled_conv_param_grads(const float* G, const float* S, const float* w, const float* b,
                                const float* gamma, int Cout, int Cin, float* dw, float* db, float* dgamma,
                                void* stream);

/* per-channel sum over N*HW of x [N,C,HW] (bias gradients); deterministic */
int64_t tdr_chansum_ws_floats(int N, int C, int HW);
int tdr_channel_sum(const float* x, int64_t x_ns, int N, int C, int HW, float* out, float* ws, void* stream);

/* strided row copy: dst[n*dst_ns + i] = src[n*src_ns + i], i < len (concat / slice glue, :719,727) */
int tdr_copy_rows(const float* src, int64_t src_ns, float* dst, int64_t dst_ns, int N, int64_t len, void* stream);
/* dst += src (same addressing) */
int tdr_add_rows(const float* src, int64_t src_ns, float* dst, int64_t dst_ns, int N, int64_t len, void* stream);
/* PixelUnshuffle(2) of a gradient: in [N,C,2H,2W] -> out [N,4C,H,W] */
int tdr_pixel_unshuffle2(const float* in, int N, int C, int H, int W, float* out, void* stream);
/* zero-pad / crop: dst[N,C,Hd,Wd] <- src[N,C,Hs,Ws] top-left aligned, zero fill (:576-585,740) */
int tdr_pad_crop(const float* src, int N, int C, int Hs, int Ws, float* dst, int Hd, int Wd, void* stream);
/* Input pipeline on the device: paired random crop (data/transforms.py:24-84) + the 8 flip / rot90 modes of
 * data_augmentation (:223-270, numpy semantics) + optional sigma-noise synthesis (restoration_dataset.py:464-476), one
 * gather per batch.  src [N][C][Hs][Ws] (per-image stride src_ns), per-sample top / left / mode int32 device arrays (NULL:
 * 0), noise [N][C][P][P] and sigma [N] (NULL: no noise / sigma 1), out [N][C][P][P].  The caller draws the parameters.
 * An image smaller than the patch (P > Hs or P > Ws) is read as if padded at the bottom / right by reflection including the
 * edge pixel -- the reference's padding() before the crop (utils/utils_image.py:243-259, cv2.BORDER_REFLECT); top / left
 * then index the padded image. */
int tdr_crop_augment(const float* src, int64_t src_ns, int N, int C, int Hs, int Ws, const int* top, const int* left,
                     const int* mode, const float* noise, const float* sigma, int P, float* out, void* stream);

/* Validation SSIM on the device -- metrics/psnr_ssim.py:131-176 (_ssim_3d: 11^3 Gaussian, sigma 1.5, over the [H,W,C]
 * volume with replicate borders; the reference runs it on the GPU too).  img1 / img2 [H][W][C] float32, C <= 4; max_value 1
 * or 255 (C1 = (0.01 max)^2, C2 = (0.03 max)^2); ws: tdr_ssim3d_ws_floats(H, W) floats; out: 1 float = mean of the SSIM map.
 * With C = 1 the channel axis drops out and the result is the 2-D replicate-border SSIM of _ssim_cly (:184-222). */
int64_t tdr_ssim3d_ws_floats(int H, int W);
int tdr_ssim3d(const float* img1, const float* img2, int H, int W, int C, float max_value, float* ws, float* out,
               void* stream);

/* The Y-channel SSIM in float64 -- metrics/psnr_ssim.py:184-222 (_ssim_cly: cv2.filter2D of float64 images with the 11 x 11 Gaussian
 * window, BORDER_REPLICATE, [0,255] constants).  img1 / img2 [H][W] float32 (the Y planes); ws: tdr_ssim_y64_ws_doubles(H, W)
 * doubles; out: 1 double = mean of the SSIM map.  Filtering, the SSIM map and its mean are evaluated in double. */
int64_t tdr_ssim_y64_ws_doubles(int H, int W);
int tdr_ssim_y64(const float* img1, const float* img2, int H, int W, double* ws, double* out, void* stream);
/* TLSC local average pooling -- `AvgPool2d.forward` of models/archs/nafnet_local_arch.py:10-75 (fast_imp = False, auto_pad):
 * out[p][y][x] = mean of the k1' x k2' box (k' = min(size, k)) whose top-left corner is (clamp(y - (H - hv) / 2, 0, hv - 1),
 * clamp(x - (W - wv) / 2, 0, wv - 1)), hv = H - k1' + 1, wv = W - k2' + 1: the valid box means replicate-padded back to H x W.
 * in / out: `planes` dense [H][W] planes; ws: tdr_local_avgpool_ws_floats(planes, H, W, k1) floats. */
int64_t tdr_local_avgpool_ws_floats(int planes, int H, int W, int k1);
int tdr_local_avgpool(const float* in, int planes, int H, int W, int k1, int k2, float* ws, float* out, void* stream);

/* ReLU backward: out = act > 0 ? go : 0 (Encoder/ResidualBlock nn.ReLU, :52,132) */
int tdr_relu_bwd(const float* go, const float* act, int64_t numel, float* out, void* stream);

/* L1 loss fwd+bwd -- losses/losses.py:11-13,52-53.  loss (1 float) = w*mean|p-t|, dpred = grad_scale*w*sign/numel.
 * grad_scale is the (power-of-two, hence exact) loss scale of the backward pass: 1 unless the fp16-split kernels carry the
 * gradient chain; the parameter gradients are divided by it again in tdr_multi_copy. */
int tdr_l1_loss(const float* pred, const float* target, int64_t numel, float loss_weight, float grad_scale,
                float* loss, float* dpred, float* ws, void* stream);
/* the same with the loss scale read from the device-resident step guard (dpred = guard->scale * w*sign/numel) */
int tdr_l1_loss_guarded(const float* pred, const float* target, int64_t numel, float loss_weight, const TdrStepGuard* guard,
                        float* loss, float* dpred, float* ws, void* stream);

/* Every pixel criterion of losses/losses.py, value + gradient in one pass (the train step takes whichever `pixel_opt.type`
 * names).  kind 0: L1Loss (:26-53), 1: MSELoss (:55-82), 2: CharbonnierLoss (:111-122: mean sqrt(d^2 + eps^2), loss_weight
 * ignored as there), 3: PSNRLoss (:84-109: w * 10/ln10 * mean_n log(mean_chw d^2 + 1e-8)), 4: PSNRLoss toY=True (BT.601 luma
 * of 3-channel images, /255).  pred / target [N][chw] dense, hw = H*W; dpred = grad_scale (* guard->scale when guard != NULL)
 * * dloss/dpred; ws: 2*1024 + 2*64*N floats. */
int tdr_pixel_loss(int kind, const float* pred, const float* target, int N, int64_t chw, int64_t hw, float loss_weight, float eps,
                   float grad_scale, const TdrStepGuard* guard, float* loss, float* dpred, float* ws, void* stream);

/* ---------------------------------------------------------------------------
 * MASA match-and-transfer (network_nafnet_guided_arch.py:495-707)
 * ------------------------------------------------------------------------- */
/* replicate-pad(1) + overlapping block cut (:627-629): feat [N,C,H,W] -> blk [N*py*px, C, ky+2, kx+2] */
int tdr_lr_blocks_fwd(const float* feat, int N, int C, int H, int W, int py, int px, int ky, int kx,
                      float* blk, void* stream);
int tdr_lr_blocks_bwd(const float* dblk, int N, int C, int H, int W, int py, int px, int ky, int kx,
                      float* dfeat, void* stream);
/* inverse L2 norms of 3x3 (dilated, zero padded by `pad`) neighbourhood patches over all channels:
 * inv[b, y, x] = 1/max(||patch||, 1e-12), output grid OHxOW (F.normalize eps, :506-507,529-530) */
int tdr_patch_inv_norm(const float* x, int B, int C, int H, int W, int OH, int OW, int dil, int pad,
                       int step, int off, float* inv, void* stream);
/* coarse arg-max (:534, :635-657): corr_sum[n,p,r] = sum_d dot_d[n,p,r]*invq_d[n,p]*invk_d[n,r];
 * index = argmax_r; box starts y1,x1 (int32) */
int tdr_coarse_argmax_box(const float* dots, const float* invq, const float* invk, int ND, int N, int P, int Hr,
                          int Wr, int diameter, int* index, int* y1, int* x1, void* stream);
/* gather ref block with python-style negative wrap (:672-678): out [N*P, C, side*s, side*s] */
int tdr_gather_ref_block(const float* feat, int N, int C, int H, int W, const int* y1, const int* x1, int P,
                         int side, int s, float* out, void* stream);
/* fine arg-max (:509-511): corr[b,p,r] = dot*invq[b,p]*invk[b,r] -> index_all[b,p] (int32), soft_att[b,p] */
int tdr_fine_argmax(const float* dots, const float* invq, const float* invk, int B, int P, int R,
                    int* index_all, float* soft_att, void* stream);
/* gradient of soft_att wrt the LR block and the ref block (through F.normalize and the selected dot) */
int tdr_fine_search_bwd(const float* datt, const float* soft_att, const int* index_all, const float* lrb,
                        const float* refb, const float* invq, const float* invk, int B, int C, int K, int D,
                        float* dlrb, float* drefb, void* stream);
/* fused transfer (:538-555 + :672-678 + :695-707): reads the ref feature map directly
 * (no block materialisation), writes the re-tiled warped map out [N,C,py*K*s,px*K*s] with n-stride out_ns. */
int tdr_transfer_fwd(const float* feat, int N, int C, int H, int W, const int* y1, const int* x1,
                     const int* index_all, const float* soft_att, int py, int px, int K, int side, int s,
                     float* out, int64_t out_ns, void* stream);
/* backward: dfeat += adjoint of the gather (caller zeroes) and datt[B,K,K] partial for this scale (+=).
 * deterministic = 0: float atomics (fast; the sum of >= 3 contributions to one address depends on arrival order, ~1e-7
 * relative run-to-run differences); 1: the scatter accumulates 64-bit fixed-point integers whose quantum follows the
 * launch's max |dout| -- integer addition is associative, the result is bit-identical from run to run (MI355X cfg2:
 * +0.5 ms per step).  ws >= tdr_transfer_ws_floats(...) 4-byte words */
int64_t tdr_transfer_ws_floats(int N, int C, int H, int W, int py, int px, int K, int s);
int tdr_transfer_bwd(const float* dout, int64_t dout_ns, const float* feat, int N, int C, int H, int W,
                     const int* y1, const int* x1, const int* index_all, const float* soft_att, int py, int px,
                     int K, int side, int s, int deterministic, float* dfeat, float* datt, float* ws, void* stream);
/* adjoint of tdr_gather_ref_block: dfeat += the ref-block gradient at every feature pixel the blocks read (wrap-aware;
 * a deterministic gather over the feature map, no atomics) */
int tdr_scatter_ref_block(const float* dblk, int N, int C, int H, int W, const int* y1, const int* x1, int P,
                          int side, float* dfeat, void* stream);

/* bit pattern of max |x| over x[n][0..per) (image stride x_ns), atomicMax-ed into *slot (caller zeroes; NaN / Inf give
 * >= 0x7f800000).  The train step's fp16-range survey: which operands of the 2-way fp16 split leave its window. */
int tdr_absmax_bits(const float* x, int64_t x_ns, int N, int64_t per, unsigned* slot, void* stream);

/* ---------------------------------------------------------------------------
 * DRSformer-ref (network_drsformer_guided_arch*.py): Top-K Sparse Attention on the MDTA machinery and the grouped
 * depthwise convolutions of its mixed-scale feed-forward.
 * tdr_tksa_softmax (:282-318): like tdr_mdta_softmax, but A = sum_m am[m] * softmax(mask_m(logits)), mask_m keeping the k4[m]
 * largest logits of a row (k = int(c/2), int(c*2/3), int(c*3/4), int(c*4/5), c = C/heads: host-computed); am = attn1..attn4.
 * tdr_tksa_bwd: dA -> W (as tdr_mdta_bwd), dtemp[heads], dam[4]; ws: 5*N*heads floats.
 * tdr_dwk_fwd/bwd (:221-247): y[n][c] = act(sum_{i<mult} w[c][i] (*) x[n][c*mult+i]), K x K (1|3|5|7), stride 1, pad K/2,
 * optional bias, mult 1 | 2 (Conv2d(2h, h, groups=h)); relu: fused ReLU (backward masks with yact = the forward output, or NULL).
 * ------------------------------------------------------------------------- */
int tdr_tksa_softmax(const float* G, const float* ss, const float* temp, const float* am, const int* k4 /*host*/, int N, int C,
                     int heads, float* A, float* AT, void* stream);
int tdr_tksa_bwd(const float* G, const float* ss, const float* temp, const float* am, const int* k4 /*host*/, const float* dA, int N,
                 int C, int heads, float* W, float* dtemp, float* dam, float* ws, void* stream);
int tdr_dwk_fwd(const float* x, int64_t x_ns, const float* w, const float* b /*[Cout] or NULL*/, int N, int Cout, int mult, int H, int W,
                int K, int dil /*1 | 2*/, int relu, float* y, int64_t y_ns, void* stream);
int64_t tdr_dwk_bwd_ws_floats(int N, int Cout, int mult, int H, int W, int K);
int tdr_dwk_bwd(const float* dy, int64_t dy_ns, const float* yact, int64_t y_ns, const float* x, int64_t x_ns, const float* w, int N,
                int Cout, int mult, int H, int W, int K, int dil, float* dx, int64_t dx_ns, float* dw, float* db /*or NULL*/,
                float* ws /*tdr_dwk_bwd_ws_floats floats*/, void* stream);
/* the same with dx += (accumulate != 0) -- the two first-stage branches of MSFN both feed project_in's output gradient; only the
 * one-pass 5x5 kernel accumulates, tdr_dwk_bwd_can_accumulate(...) != 0 tells whether the arguments take it */
int tdr_dwk_bwd_can_accumulate(int W, int K, int dil, int64_t dy_ns, int64_t y_ns, int64_t x_ns, int64_t dx_ns);
int tdr_dwk_bwd_acc(const float* dy, int64_t dy_ns, const float* yact, int64_t y_ns, const float* x, int64_t x_ns, const float* w,
                    int N, int Cout, int mult, int H, int W, int K, int dil, float* dx, int64_t dx_ns, int accumulate, float* dw,
                    float* db, float* ws, void* stream);
/* MEFC sub-network pieces (network_drsformer_guided_arch.py:371-548): the 3x3 average pool with count_include_pad=False
 * (adjoint = 1: its backward), the gating MLP's Linears (tiny: N x Cin -> Cout, one wave per output), row softmax (dy != NULL:
 * backward from the softmax output), dst[n] = src[n] * w[n * w_stride] (the per-image operation weights), per-image dot
 * products (their gradients; ws: 64 * N floats). */
int tdr_avgpool3(const float* in, int planes, int H, int W, int adjoint, float* out, void* stream);
int tdr_add_relu(const float* a, const float* b, int64_t numel, float* out, void* stream);      /* relu(a + b) (:407) */
int tdr_linear_small_fwd(const float* x, const float* W, const float* b, int N, int Cin, int Cout, int relu, float* y, void* stream);
int tdr_linear_small_bwd(const float* dy, const float* yact /*or NULL*/, const float* x, const float* W, int N, int Cin, int Cout,
                         float* dx, float* dW, float* db, void* stream);
int tdr_softmax_rows(const float* x, const float* dy /*NULL: forward*/, int rows, int L, float* out, void* stream);
int tdr_scale_copy(const float* src, int64_t src_ns, const float* w, int w_stride, int N, int64_t len, float* dst, int64_t dst_ns,
                   void* stream);
int tdr_rows_dot(const float* a, int64_t a_ns, const float* b, int64_t b_ns, int N, int64_t len, float* out, int out_stride, float* ws,
                 void* stream);

/* ---------------------------------------------------------------------------
 * PromptIR-ref PromptGenBlock (network_promptir_guided_arch.py:417-441), everything around its 3x3 convolution:
 *   emb = mean_hw x (tdr_plane_mean);  w = softmax(Linear(emb)) (tdr_prompt_weights_*);
 *   prompt[n] = sum_k w[n][k] * P[k] (tdr_prompt_mix_*), P = the L parameter planes after the bilinear resize to (H, W)
 *   (tdr_resize_bilinear / tdr_resize_bilinear_bwd: weighted sum and resize commute, the resize runs once per step).
 * prompt_len L <= 16.  All reductions fixed-order.
 * ------------------------------------------------------------------------- */
int tdr_plane_mean(const float* x, int64_t x_ns, int N, int C, int HW, float* out /*[N][C]*/, void* stream);
/* x[n][c][:] += v[n][c] * scale   (gradient of the mean: scale = 1/HW) */
int tdr_plane_add(float* x, int64_t x_ns, const float* v, float scale, int N, int C, int HW, void* stream);
int tdr_prompt_weights_fwd(const float* emb, const float* W /*[L][C]*/, const float* b /*[L] or NULL*/, int N, int C, int L,
                           float* w /*[N][L]*/, void* stream);
int tdr_prompt_weights_bwd(const float* emb, const float* W, const float* w, const float* dw, int N, int C, int L,
                           float* dW, float* db /*or NULL*/, float* demb /*[N][C]*/, void* stream);
int tdr_prompt_mix_fwd(const float* w, const float* P /*[L][E]*/, int N, int L, int64_t E, float* out /*[N][E]*/, void* stream);
int64_t tdr_prompt_mix_bwd_ws_floats(int N, int L);
int tdr_prompt_mix_bwd(const float* w, const float* P, const float* d /*[N][E]*/, int N, int L, int64_t E, float* dP /*[L][E]*/,
                       float* dw /*[N][L]*/, float* ws, void* stream);
/* adjoint of tdr_resize_bilinear: ddst [planes][Hd][Wd] -> dsrc [planes][Hs][Ws] (gather, no atomics) */
int tdr_resize_bilinear_bwd(const float* ddst, int planes, int Hs, int Ws, int Hd, int Wd, float* dsrc, void* stream);

/* ---------------------------------------------------------------------------
 * Frozen ViT window matcher (DINOv2 ViT-B/14, forward only): models/image_restoration_ref_model.py:215-247,
 * models/dino/.  Activations are channel-major [B][D][T]: every Linear is tdr_conv_forward (1x1), the token
 * LayerNorm is tdr_layernorm2d_fwd (eps 1e-6), the MLP GELU is the relu=2 epilogue.
 * ------------------------------------------------------------------------- */
/* F.interpolate(bilinear, align_corners=False) of `planes` = B*C planes */
int tdr_resize_bilinear(const float* src, int planes, int Hs, int Ws, float* dst, int Hd, int Wd, void* stream);
/* F.unfold(ref,(h,h),stride) -> [B*N][C][h][h], window n = wy*nx + wx (:220-224) */
int tdr_unfold_windows(const float* ref, int B, int C, int Hr, int Wr, int h, int stride, float* out, void* stream);
/* Token tensors use a padded row length LD (multiple of 32, >= 1+T): column 0 class token, 1..T patches, rest padding,
 * so the 1x1-conv kernels see [B][D][LD] as an [LD/32] x 32 image.
 * `flat` != 0 selects the BATCH-FLATTENED layout [D][B*LD] instead (image b's tokens at columns b*LD .. b*LD+LD-1 of one pixel
 * axis): every Linear of the encoder is then ONE GEMM over B*LD pixels (full 128-pixel tiles, one pass over the weights)
 * instead of B GEMMs over LD = 288 -- what the stage-A CLIP encoder uses (main_train_i2t_mapping.py:726-731).
 * p x p stride-p patch gather: x [B][Ci][H][W] -> [B][Ci*p*p][LD], patch t at column 1+t (patch_embed.py:26-80) */
int tdr_patchify(const float* x, int B, int Ci, int H, int W, int p, int LD, int flat, float* out, void* stream);
/* in place: column 0 = cls + pos[:,0], columns 1..T += pos, padding = 0 (pos channel-major [D][1+T]; vision_transformers.py:209-221) */
int tdr_vit_assemble(float* tok, const float* cls, const float* pos, int B, int D, int T, int LD, int flat, void* stream);
/* multi-head softmax(q k^T scale) v over the first T columns; qkv [B][3C][LD] -> out [B][C][LD] (padding columns zeroed);
 * head dim C/heads in {16,32,64} (attention.py:56-71) */
int tdr_attention_fwd(const float* qkv, int B, int C, int heads, int T, int LD, float scale, float* out, void* stream);
/* the same with the arithmetic named: math 0 = exact fp32 MFMA (what tdr_attention_fwd runs), 1 = 3-way bf16 split (q, k, v and P as three bf16
 * planes, 6 bf16 MFMA products per fp32 product, fp32 accumulate and softmax: the default arithmetic, any fp32 exponent), 2 = 2-way fp16 split (3 f16 MFMA
 * products per fp32 product, fp32 accumulate and softmax) for the frozen no-grad ViTs -- q, k, v must lie in the fp16 range;
 * 3 = plain fp16 MFMA (one product, reduced precision) for the DINOv2 window matcher only: its sole output is an arg-max */
int tdr_attention_fwd_math(const float* qkv, int B, int C, int heads, int T, int LD, float scale, int math, int flat,
                           float* out, void* stream);
/* cosine similarity of flattened patch-token maps (columns 1..T1-1), first arg-max, window gather:
 * fl [B][D][LD], fr [B*N][D][LD], windows [B*N][per] -> corr [B][N], index [B] (int32), ref_in [B][per] (:230-243) */
int tdr_token_match(const float* fl, const float* fr, const float* windows, int B, int N, int D, int T1, int LD, int64_t per,
                    float* corr, int* index, float* ref_in, void* stream);

/* Token-major fp16 pipeline of the frozen DINOv2 matcher (csrc/tdr_tok16.hip; replaces the per-block calls of
 * models/dino/vision_transformers.py inside get_ref_in, image_restoration_ref_model.py:215-247, whose only output is an arg-max).
 * The residual stream is fp32 [P][D] (P = images x padded tokens per image, D contiguous); every GEMM operand is fp16 [P][K],
 * rounded once by its producer -- the arithmetic of tdr_attention_fwd_math code 3 / the single-product Linears, summation order
 * aside.  Pointers named *16 are device arrays of IEEE binary16. */
/* dst[b][c][r] = src[b][r][c] (contiguous fp32 batches): entering / leaving the channel-major layout of tdr_vit_assemble and
 * tdr_token_match */
int tdr_transpose_f32(const float* src, int batch, int R, int C, float* dst, void* stream);
/* nn.LayerNorm(D) over each of the P token rows (D % 4 == 0, D <= 1280); out_f16 0: fp32 [P][D]; 1: fp16 [P][D];
 * 2: the 2-way split, hi | lo fp16 planes [2][P][D] (hi = fp16(y), lo = fp16(y - hi));
 * 3: the 3-way split, h | m | l bf16 planes [3][P][D] (h = bf16(y), m = bf16(y - h), l = bf16(y - h - m): y = h + m + l exactly) */
int tdr_tok_layernorm(const float* x, const float* w, const float* b, int64_t P, int D, float eps, int out_f16, void* out,
                      void* stream);
/* nn.Linear: acc = x16 [P][K] . w16 [N][K]^T in fp32 (N % 128 == 0, K % 64 == 0); bias may be NULL.
 * epi 0: y16 [P][N] = acc + bias;  1: y16 = erf-GELU(acc + bias);  2: res [P][N] (fp32, in place) += ls[n] * (acc + bias)
 * (LayerScale + residual; ls NULL = 1) */
int tdr_tok16_gemm(const void* x16, const void* w16, const float* bias, int64_t P, int N, int K, int epi, void* y16, float* res,
                   const float* ls, void* stream);
/* The same GEMM on 2-way split operands (fp32-faithful: the frozen CLIP encoder of the stage-A trainers,
 * scripts/train/main_train_i2t_mapping.py:564,726-731, feeds a trained path): x16x2 [2][P][K] and w16x2 [2][N][K] are hi | lo planes,
 * acc = x_lo w_hi + x_hi w_lo + x_hi w_hi in fp32 (N % 128 == 0, K % 32 == 0, P % 8 == 0); bias may be NULL.
 * epi 2: out32 [P][N] (in place) += acc + bias;  3: out32 [N][P] = acc + bias (channel-major: the layout of tdr_attention_fwd_math);
 * epi 4: y16x2 [2][P][N] = split(act(acc + bias)), act 0 none / 2 erf-GELU / 3 quick_gelu (the codes of tdr_conv_forward) */
int tdr_tok16x2_gemm(const void* x16x2, const void* w16x2, const float* bias, int64_t P, int N, int K, int epi, int act, void* y16x2,
                     float* out32, void* stream);
/* fp32 channel-major src [C][P] -> token-major hi | lo planes dst16x2 [2][P][C] */
int tdr_cm_to_tok16x2(const float* src, int C, int64_t P, void* dst16x2, void* stream);
/* The same GEMM on 3-way split operands (the library's default arithmetic, TDR_MATH=bx3: x = h + m + l in bf16, exactly; six products
 * per fragment pair, fp32 accumulate; any fp32 exponent): the frozen DINOv2 matcher (models/dino/block.py:42-113, attention.py:36-71)
 * at its default arithmetic.  x16x3 [3][P][K], w16x3 [3][N][K] bf16 planes; epi / act as tdr_tok16x2_gemm (y16x3 [3][P][N]);
 * ls (may be NULL): epi 2 adds ls[n] * (acc + bias) -- the LayerScale of the block's residual branches */
int tdr_tok16x3_gemm(const void* x16x3, const void* w16x3, const float* bias, int64_t P, int N, int K, int epi, int act, void* y16x3,
                     float* out32, const float* ls, void* stream);
/* fp32 channel-major src [C][P] -> token-major h | m | l bf16 planes dst16x3 [3][P][C] */
int tdr_cm_to_tok16x3(const float* src, int C, int64_t P, void* dst16x3, void* stream);
/* softmax(q k^T * scale) v per head over the first T rows of each image: qkv16 [B][LD][3C] (q | k | v column blocks, head-major
 * inside each, head dim 64), out16 [B][LD][C]; rows T..LD-1 of out16 are zero */
int tdr_tok16_attention(const void* qkv16, int B, int C, int heads, int T, int LD, float scale, void* out16, void* stream);

/* ---------------------------------------------------------------------------
 * Stage-A (image-to-text mapping) glue: scripts/train/main_train_i2t_mapping.py:40-81 (Mapper), :85-98,197-233 (injected
 * cross-attention).  The frozen CLIP ViT image encoder (:564,726-731) reuses the ViT kernels above (patchify, assemble,
 * tdr_attention_fwd incl. head dim 80, quick_gelu = relu code 3 of tdr_conv_forward).  Linears are 1x1 convs over
 * channel-major tokens [B][D][LD]; nn.LayerNorm is tdr_layernorm2d_* (eps 1e-5).
 * ------------------------------------------------------------------------- */
/* nn.LeakyReLU (slope > 0); the backward takes the activation OUTPUT y (same sign as the input) */
int tdr_leaky_relu_fwd(const float* x, int64_t numel, float slope, float* y, void* stream);
int tdr_leaky_relu_bwd(const float* go, const float* y, int64_t numel, float slope, float* gx, void* stream);
/* dst [D][32]: dst[d][b] = src[b][d][col] (b < B <= 32), zero-padded: the class tokens `embs[:, :1]` as one 32-pixel row (:77) */
int tdr_gather_col(const float* src, int B, int D, int LD, int col, float* dst, void* stream);
/* out[b][word][d] = cls[d][b] + mean_{t=1..T} patch[b][d][t]   (:77-79; out [B][words][D]) */
int tdr_mapper_combine(const float* cls, const float* patch, int B, int D, int LD, int T, int words, int word, float* out,
                       void* stream);
/* gradient of the above w.r.t. cls [D][32] and patch [B][D][LD] (go [B][words][D]) */
int tdr_mapper_combine_bwd(const float* go, int B, int D, int LD, int T, int words, int word, float* dcls, float* dpatch,
                           void* stream);
/* softmax(Q K^T * scale) V with separate tensors (inj_forward_crossattention :216-225): q [B][C][LDq] (Tq valid columns),
 * k, v [B][C][LDk] (Tk valid), channel-major, heads split along C (head dim C/heads in {16,32,64,80}) -> out [B][C][LDq]
 * (padding columns zeroed); lse [B][heads][LDq] (may be NULL) = log-sum-exp of the scaled scores, kept for the backward */
int tdr_cross_attention_fwd(const float* q, const float* k, const float* v, int B, int C, int heads, int Tq, int LDq,
                            int Tk, int LDk, float scale, float* out, float* lse, void* stream);
/* gradient of the above: dout [B][C][LDq] -> dq [B][C][LDq], dk, dv [B][C][LDk] (padding columns zeroed).
 * Two deterministic passes (per-query-tile dq, per-key-tile dk/dv), no atomics.  ws >= B*heads*LDq floats.
 * dq may be NULL (queries from a frozen producer: the stage-A step only needs dk / dv): the dq pass is skipped. */
int tdr_cross_attention_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                            const float* lse, int B, int C, int heads, int Tq, int LDq, int Tk, int LDk, float scale,
                            float* dq, float* dk, float* dv, float* ws, void* stream);
/* dst[b][c][r] = src[b][r][c] (r < R), 0 for R <= r < LDd: token-major [B][T][D] <-> channel-major [B][D][LD] */
int tdr_transpose_pad(const float* src, int B, int R, int C, int LDd, float* dst, void* stream);
/* Grouped Mapper (:40-81): the 2 x num_words independent MLPs run as G-way grouped GEMMs (tdr_conv_forward / tdr_conv_wgrad with
 * N = G "images", per-image weights wp_ns and biases bias_ns, in_ns = 0 for the shared first-layer input) over tensors
 * [G][C][P] (P = the tokens of ALL images along one axis).  What the GEMMs leave:
 * nn.LayerNorm(C, eps) + nn.LeakyReLU(slope) fused, per-word affine parameters w, b [G][C]; mu, rstd [G][P] saved */
int tdr_group_ln_act_fwd(const float* z, const float* w, const float* b, float eps, float slope, int G, int C, int P,
                         float* y, float* mu, float* rstd, void* stream);
/* its backward from dy (w.r.t. the activation output) and y (that output): dz [G][C][P], gw, gb [G][C], and gs [G][C] = the
 * pixel sums of dz (the bias gradient of the Linear that produced z) -- fixed-order two-stage sums; ws >= tdr_group_ln_ws_floats */
int64_t tdr_group_ln_ws_floats(int G, int C, int P);
int tdr_group_ln_act_bwd(const float* dy, const float* y, const float* z, const float* mu, const float* rstd, const float* w,
                         float slope, int G, int C, int P, float* dz, float* gw, float* gb, float* gs, float* ws, void* stream);
/* out[b][g][d] = cls[g][d][b] + mean_{t=1..T} patch[g][d][b*LD + t] for all G words at once (:77-79; cls [G][D][32] holds the B
 * class tokens as one 32-pixel row, patch [G][D][B*LD], out [B][G][D]), and its gradient */
int tdr_mapper_combine_all(const float* cls, const float* patch, int B, int G, int D, int LD, int T, float* out, void* stream);
int tdr_mapper_combine_all_bwd(const float* go, int B, int G, int D, int LD, int T, float* dcls, float* dpatch,
                               float* gsum /* [G][D] = sum_b go: the pixel sum of dcls and of dpatch */, void* stream);
/* tdr_gather_col for any token layout: dst[d][b] = src[b*img_stride + d*ch_stride + col] */
int tdr_gather_col_strided(const float* src, int B, int D, int64_t img_stride, int64_t ch_stride, int col, float* dst, void* stream);
/* Split-K for the single-round long-K Linears of the frozen ViTs (1280 -> 1280 and 5120 -> 1280 over ~1 000 tokens: 180 workgroups on
 * 256 CUs, each walking 20 - 80 K stages).  The K chunks run as the N = S "images" of ONE tdr_conv_forward launch (input viewed as
 * [S][K/S][P], per-image packed weights wp_ns) into partial sums part [S][C][P]; this finishes them:
 *   out[c][p] = act((sum_s part[s][c][p] + bias[c]) * scale[c] + res[c][p])   (tdr_conv_forward's STD epilogue order; bias / scale /
 * res may be NULL; relu codes as TdrConvDesc.relu), fixed summation order. */
int tdr_splitk_finish(const float* part, int S, int C, int64_t P, const float* bias, const float* scale, const float* res, int relu,
                      float* out, void* stream);
/* Stage-A train step glue (main_train_i2t_mapping.py:704-760).
 * inj_forward_text's embedding injection (:139-151) + position embedding, written channel-major [B][D][LD]: the L mapper words
 * inj [B][L][D] replace the placeholder token at position idx[b] of the prompt ids [B][S] (int32), the rest of the prompt moves
 * up by L - 1, what no longer fits in S positions is dropped; tok_emb [V][D], pos_emb [S][D]; columns S..LD-1 are zero. */
int tdr_text_inject_fwd(const int* ids, const float* tok_emb, const float* pos_emb, const float* inj, const int* idx,
                        int B, int S, int D, int L, int LD, float* out, void* stream);
/* its gradient w.r.t. inj: dinj[b][j][:] = dnew[b][:][idx[b] + j] (0 past the end of the prompt) */
int tdr_text_inject_bwd(const float* dnew, const int* idx, int B, int S, int D, int L, int LD, float* dinj, void* stream);
/* DDIMScheduler.add_noise (:717; diffusers, third party): out = sqrt(ac[t_b]) x + sqrt(1 - ac[t_b]) noise, t [B] int32 */
int tdr_add_noise(const float* x, const float* noise, const int* t, const float* alphas_cumprod, int B, int64_t per,
                  float* out, void* stream);
/* The stand-in for the frozen SD UNet (third party, absent: SURVEY 8d cfg4 "fixed random linear stub") needs three pieces of
 * glue around the real injected cross-attention: the level input [B][C+4][H/f][W/f] = f x f average pool of x [B][C][H][W]
 * followed by 4 planes of time features sin / cos(2 pi k t_b / 1000), k = 1, 2; nearest-neighbour upsampling by f
 * (optionally accumulated into dst); and its adjoint, the f x f block sum. */
int tdr_pool_time(const float* x, const int* t, int B, int C, int H, int W, int f, float* out, void* stream);
int tdr_upsample_nearest_add(const float* src, int planes, int H, int W, int f, int accumulate, float* dst, void* stream);
int tdr_pool_sum(const float* src, int planes, int H, int W, int f, float* dst, void* stream);

/* ---------------------------------------------------------------------------
 * Optimiser: global-norm clip (max_norm 0.01) + AdamW, multi-tensor
 *   models/image_restoration_ref_model.py:172-178,276-279
 * ------------------------------------------------------------------------- */
/* Work is cut into tdr_optim_chunk()-element chunks by a host-built table:
 * chunk k covers elements [chunk_index[k]*CHUNK, +CHUNK) of tensor chunk_tensor[k].
 * grads/params/...: device arrays of n_tensors device pointers; sizes: device int64[n_tensors].
 * sumsq[0] = sum g^2 over all tensors (double, deterministic two-stage; partial: n_chunks doubles). */
int tdr_optim_chunk(void);
/* dst[t][0..sizes[t]) = scale * src[t][..] for all tensors of the chunk table, one launch (gradient arena gather;
 * scale = 1 / loss scale) */
int tdr_multi_copy(const float* const* src, float* const* dst, const int64_t* sizes, const int* chunk_tensor,
                   const int* chunk_index, int n_chunks, float scale, void* stream);
int tdr_grad_sumsq(const float* const* grads, const int64_t* sizes, const int* chunk_tensor, const int* chunk_index,
                   int n_chunks, double* partial, double* sumsq, void* stream);
/* p,m,v updated in place.  coef = min(1, max_norm/(sqrt(sumsq)+1e-6)) computed on device when
 * use_clip (torch.nn.utils.clip_grad_norm_); lr = group_lr[group[t]] (host array, <= 4 groups);
 * torch.optim.AdamW update (decoupled decay, bias correction by `step`). */
int tdr_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const int64_t* sizes, const int* group, const int* chunk_tensor, const int* chunk_index, int n_chunks,
                   const double* sumsq, const float* group_lr, int n_groups, float max_norm, int use_clip, float beta1,
                   float beta2, float eps, float weight_decay, int step, void* stream);
/* Same update, per-step scalars in device memory: hp = {lr[0..3], 1-beta1^step, sqrt(1-beta2^step)} (6 floats).
 * Launch arguments are step-invariant, so the optimiser can be replayed from a captured hipGraph. */
int tdr_adamw_step_dev(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const int64_t* sizes, const int* group, const int* chunk_tensor, const int* chunk_index, int n_chunks,
                       const double* sumsq, const float* hp, float max_norm, int use_clip, float beta1, float beta2,
                       float eps, float weight_decay, void* stream);
/* ---------------------------------------------------------------------------
 * Fused second half of a NAFBlock (network_nafnet_guided_arch.py:226-238), one launch instead of four:
 *   y = x + conv3(g * sca) * beta;  yn = norm2(y);  t4 = conv4(yn);  out = y + conv5(t4[:, :C] * t4[:, C:]) * gamma
 * One workgroup owns 64 pixels x all channels; every tensor the backward pass keeps (y, mu, rs, yn, t4) is an output.
 * w3/w4/w5: tdr_pack_weights_hx2(mode FWD) of conv3 (C x C), conv4 (2C x C), conv5 (C x C); w_fmt must be 2.
 * Supported: tdr_naf_tail_supported(C, HW): C in {32, 64, 128, 256} (C / 32 waves per workgroup), HW % 64 == 0. */
typedef struct TdrNafTailDesc {
    int N, C, HW, w_fmt;
    int c_out;                               /* rows of conv5 produced: C, or C / 2 (the `[:, :chan]` slice of a fusion block); 0 = C */
    float eps;
    const float* g;   int64_t g_ns;          /* [N, C, HW] SimpleGate output of the first half */
    const float* sca;                        /* [N, C] channel attention */
    const float* x;   int64_t x_ns;          /* block input (residual) */
    const void *w3, *w4, *w5;
    const float *b3, *beta, *lnw, *lnb, *b4, *b5, *gamma;
    float* y;   int64_t y_ns;
    float *mu, *rs;                          /* [N, HW] LayerNorm statistics of y */
    float* yn;  int64_t yn_ns;
    float* t4;  int64_t t4_ns;               /* [N, 2C, HW] */
    float* out; int64_t out_ns;
} TdrNafTailDesc;
int tdr_naf_tail_supported(int C, int HW);
int tdr_naf_tail_fwd(const TdrNafTailDesc* d, void* stream);
/* Data-gradient chain of the same half (autograd of :230-238), one launch instead of three:
 *   dt4 = SimpleGate'(W5^T (dout * gamma); t4);  dyn = W4^T dt4;  dy = LayerNorm2d'(dyn; y, mu, rs, lnw) + dout
 * plus the LayerNorm parameter gradients gw = sum dyn * yhat, gb = sum dyn (per-workgroup partials in ws, reduced in a
 * fixed order).  dt4 is an output because conv4's weight gradient (tdr_conv_wgrad) reads it.
 * w5t / w4t: tdr_pack_weights_hx2(mode DGRAD_S1) of conv5 / conv4.  ws: tdr_naf_tail_bwd_ws_floats(N, C, HW) floats.
 * gw = gb = NULL (tail and head backward alike): the per-workgroup partials [N * HW / 64][2][C] are left at the start of ws and the
 * caller finishes them later with tdr_pair_sum_partials(ws, N * HW / 64, C, gw, gb, ws + N * HW / 64 * 2 * C) -- the LayerNorm
 * parameter gradients are leaves, so a trainer may run that reduction off the data-gradient chain. */
typedef struct TdrNafTailBwdDesc {
    int N, C, HW, w_fmt;
    int c_out;                               /* channels of dout (C or C / 2; 0 = C) */
    const float* dout; int64_t dout_ns;
    const float* gamma;
    const float* t4;   int64_t t4_ns;
    const float* y;    int64_t y_ns;
    const float *mu, *rs, *lnw;
    const void *w5t, *w4t;
    float* dt4; int64_t dt4_ns;
    float* dy;  int64_t dy_ns;
    float *gw, *gb;
    float* ws;
    /* optional conv3 data-gradient stage in the same launch (NULL w3t = off): dgp = sca[n] * (W3^T (beta * dy)), i.e. the
     * gradient of g through `y = inp + conv3(g * sca) * beta` (:226-230) WITHOUT the pooled-gradient term of the SCA branch,
     * which tdr_dwsg_bwd_biased adds per (image, channel) plane.  w3t: tdr_pack_weights_hx2(mode DGRAD_S1) of conv3. */
    const void* w3t; const float* beta; const float* sca /*[N, C]*/;
    float* dgp; int64_t dgp_ns;
} TdrNafTailBwdDesc;
int64_t tdr_naf_tail_bwd_ws_floats(int N, int C, int HW);
int tdr_naf_tail_bwd(const TdrNafTailBwdDesc* d, void* stream);
/* First half of a NAFBlock up to the depthwise conv (network_nafnet_guided_arch.py:216-219), one launch:
 *   xn = LayerNorm2d(x; lnw, lnb, eps)   (mu, rs kept for the backward pass)        t1 = conv1(xn) + b1
 * w1: tdr_pack_weights_hx2(mode FWD) of conv1 (2C x C).  Same support as tdr_naf_tail_fwd. */
typedef struct TdrNafHeadFwdDesc {
    int N, C, HW, w_fmt;
    const float* x; int64_t x_ns;
    const float *lnw, *lnb; float eps;
    const void* w1; const float* b1;
    float *mu, *rs;                          /* [N, HW] */
    float* xn; int64_t xn_ns;                /* [N, C, HW] */
    float* t1; int64_t t1_ns;                /* [N, 2C, HW] */
} TdrNafHeadFwdDesc;
int tdr_naf_head_fwd(const TdrNafHeadFwdDesc* d, void* stream);
/* The first half's data gradients (autograd of :216-225 from conv1 back), one launch instead of two:
 *   dxn = W1^T dt1;  dx = LayerNorm2d'(dxn; x, mu, rs, lnw) + res        (res = gradient of the `inp + ...` skip)
 * plus norm1's parameter gradients.  w1t: tdr_pack_weights_hx2(mode DGRAD_S1) of conv1 (2C x C).  Same support / ws. */
typedef struct TdrNafHeadBwdDesc {
    int N, C, HW, w_fmt;
    const float* dt1; int64_t dt1_ns;        /* [N, 2C, HW] gradient of conv1's output */
    const float* x;   int64_t x_ns;          /* block input */
    const float *mu, *rs, *lnw;
    const void* w1t;
    const float* res; int64_t res_ns;
    float* dx; int64_t dx_ns;
    float *gw, *gb;
    float* ws;
} TdrNafHeadBwdDesc;
int tdr_naf_head_bwd(const TdrNafHeadBwdDesc* d, void* stream);
/* part [nparts][2][C] -> o0[c] = sum_k part[k][0][c], o1[c] = sum_k part[k][1][c], fixed summation order.
 * mid: scratch of tdr_pair_sum_mid_floats(nparts, C) floats (a 256-to-1 first stage above 1024 rows; may be NULL when 0) */
int64_t tdr_pair_sum_mid_floats(int nparts, int C);
int tdr_pair_sum_partials(const float* part, int nparts, int C, float* o0, float* o1, float* mid, void* stream);
/* Table-driven forms of three small finishing reductions (ABI 107): the deferred leaves of the NAFBlock levels are dozens of small problems
 * (28 blocks at the 64x64 level alone: 56 LayerNorm-partial reductions, 28 depthwise parameter finishes, 28 conv5 / gamma parameter
 * gradients), each a 6 us launch; one launch per kind takes a table of 64-bit words in DEVICE memory instead -- device pointers followed by the
 * problem's shape, so problems of different shapes share the launch (the grid is sized for the widest, `max_*`).  Per problem the summation
 * order of the single-problem entry point: bit-identical results.
 *   tdr_pair_sum_partials_multi:       rows {part, o0, o1, nparts, C};  nparts <= 1024 (the one-stage reduction)
 *   tdr_dw_param_finish_multi:         rows {ws, dw, db, N, C, nb};     nb = tdr_dw_param_finish_nb(H, W)
 *   tdr_scaled_conv_param_grads_multi: rows {G, S, w, b, gamma, dw, db, dgamma, Cout, Cin} */
int tdr_pair_sum_partials_multi(const void* table, int nprob, int max_C, void* stream);
int tdr_dw_param_finish_nb(int H, int W);
int tdr_dw_param_finish_multi(const void* table, int nprob, int max_C, void* stream);
int tdr_scaled_conv_param_grads_multi(const void* table, int nprob, int max_Cout, void* stream);

/* ---------------------------------------------------------------------------
 * Data-parallel exchange over RCCL / xGMI: replaces DistributedDataParallel's gradient all-reduce and
 * constructor broadcast (models/base_model.py:76-82) and reduce_loss_dict's dist.reduce (:361-372), i.e. what the
 * reference reaches through torch.distributed.launch + utils/utils_dist.py:10-83.  One process per GPU; the
 * communicator binds to the caller's current HIP device.  Rendezvous: rank 0 calls tdr_comm_unique_id and hands the
 * tdr_comm_unique_id_bytes() (=128) bytes to every rank by any side channel (env/TCP store/file), then ALL ranks call
 * tdr_comm_init.  Collectives are in place on fp32 device buffers, enqueued on `stream`, capturable in a hipGraph.
 * librccl.so.1 is bound at run time (the copy already resident in the process, else /opt/rocm/lib). */
typedef struct TdrComm TdrComm;
int tdr_comm_available(void);          /* 1 when librccl.so.1 and every entry point used here resolve in this process, else 0 (no side effect) */
int tdr_comm_unique_id_bytes(void);
int tdr_comm_unique_id(void* id_out);
int tdr_comm_init(TdrComm** comm, int rank, int world, const void* unique_id);
int tdr_comm_allreduce(TdrComm* comm, float* buf, int64_t count, int average, void* stream);  /* sum, or mean over ranks */
int tdr_comm_reduce(TdrComm* comm, float* buf, int64_t count, int root, void* stream);        /* sum to `root` */
int tdr_comm_broadcast(TdrComm* comm, float* buf, int64_t count, int root, void* stream);
int tdr_comm_rank(const TdrComm* comm);
int tdr_comm_world(const TdrComm* comm);
int tdr_comm_destroy(TdrComm* comm);

/* EMA of the weights, models/base_model.py:54-62 (`p_ema.mul_(decay).add_(p, alpha=1-decay)` per tensor): one launch
 * over the same kind of (tensor, chunk) table. */
int tdr_multi_ema(const float* const* src, float* const* dst, const int64_t* sizes, const int* chunk_tensor,
                  const int* chunk_index, int n_chunks, float decay, void* stream);
/* Guarded step (TdrStepGuard above): gather with scale = guard->inv_scale; gradient norm over the tensors with
 * group[t] >= 0 (group[t] < 0 = frozen: the reference's requires_grad_(False) on the "masa" parameters while
 * current_iter < fix_iterations, image_restoration_ref_model.py:205-212) + verdict / step count / bias corrections /
 * loss-scale update; AdamW (coupled_decay 0) or torch.optim.Adam (coupled_decay 1: weight_decay * p added to the
 * gradient, :176-178) update that is skipped when the verdict is "not finite".  hp = {lr[0..3]} in device memory. */
int tdr_multi_copy_guarded(const float* const* src, float* const* dst, const int64_t* sizes, const int* chunk_tensor,
                           const int* chunk_index, int n_chunks, const TdrStepGuard* guard, void* stream);
int tdr_grad_sumsq_guarded(const float* const* grads, const int64_t* sizes, const int* group, const int* chunk_tensor,
                           const int* chunk_index, int n_chunks, double* partial, double* sumsq, TdrStepGuard* guard,
                           float beta1, float beta2, void* stream);
int tdr_adamw_step_guarded(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                           const int64_t* sizes, const int* group, const int* chunk_tensor, const int* chunk_index,
                           int n_chunks, const double* sumsq, const float* hp, const TdrStepGuard* guard, float max_norm,
                           int use_clip, int coupled_decay, float beta1, float beta2, float eps, float weight_decay,
                           void* stream);

/* ---------------------------------------------------------------------------
 * Un-guided SFNet (models/archs/network_sfnet_guided_arch.py:320-407, sfnet_arch_utils.py:76-265), training-mode
 * semantics.  The convolutions run on tdr_conv_forward / tdr_conv_wgrad; these are the operators around them (csrc/tdr_sfnet.hip).
 * --------------------------------------------------------------------------- */
/* exact-erf GELU after a convolution (BasicConv, sfnet_arch_utils.py:76-98): z = x + bias[channel] (bias may be NULL; C / HW give the
 * channel of element i as (i / HW) % C), y = gelu(z); z_out (may be NULL, may alias x) receives z for the backward pass: dz = dy * gelu'(z) */
int tdr_gelu_fwd(const float* x, const float* bias, int C, int HW, float* z_out, float* y, int64_t n, void* stream);
int tdr_gelu_bwd(const float* dy, const float* z, float* dz, int64_t n, void* stream);
/* F.interpolate(scale_factor=0.5) in its default nearest mode (:368-369): y[., i, j] = x[., 2i, 2j]; planes = N * C */
int tdr_subsample2(const float* x, int planes, int H, int W, float* y, void* stream);
/* InstanceNorm2d(C, affine=True) (SCM, :208): per (image, channel) plane; mu / rs [N * C] kept for the backward pass.  ws: 2 N C floats */
int tdr_instnorm_fwd(const float* x, const float* w, const float* b, float eps, int N, int C, int HW, float* y, float* mu, float* rs, void* stream);
int tdr_instnorm_bwd(const float* dy, const float* x, const float* mu, const float* rs, const float* w, int N, int C, int HW, float* dx,
                     float* dw, float* db, float* ws, void* stream);
/* Gap (sfnet_arch_utils.py:101-117; q = 1, shift = 1: ph = fscale_h, pl = fscale_d) and Patch_ap (:239-265; q = 2: the four quadrants
 * of the plane, shift = 0: ph = h, pl = l), parameters [C * q * q] in the reference's `(c p1 p2)` order:
 *   y = x * A + mean_region(x) * B,   A = ph + shift,  B = pl - A
 * on channel slices addressed by an image stride; mean [N][C][q * q] is kept.  Backward: dx and the parameter gradients dph, dpl.
 * ws: 2 N C q q floats */
int tdr_region_affine_fwd(const float* x, int64_t x_ns, const float* ph, const float* pl, float shift, int q, int N, int C, int H, int W,
                          float* y, int64_t y_ns, float* mean, void* stream);
int tdr_region_affine_bwd(const float* dy, int64_t dy_ns, const float* x, int64_t x_ns, const float* ph, const float* pl, float shift,
                          const float* mean, int q, int N, int C, int H, int W, float* dx, int64_t dx_ns, float* dph, float* dpl, float* ws,
                          void* stream);
/* (the pooled vectors come from tdr_plane_mean above) */
/* dynamic_filter (:152-192) + SFconv (:195-236) on the pooled vectors ap [N][c], one workgroup:
 *   taps [N][G * KK] = softmax over the KK = k * k taps of each (image, group) of BatchNorm2d_train(conv1x1(ap)) -- the running buffers and
 *   num_batches_tracked are moved in place --;  [ah ; al] [N][c] each = softmax over all 2c entries of [fcs0(fc(ap)) ; fcs1(fc(ap))].
 * xhat [N][GK], rstd [GK], z [N][d], att [N][2c] are what the backward pass needs. */
typedef struct TdrSfDynVecDesc {
    int N, c, GK, KK, d;
    float eps, momentum;
    const float *ap, *wconv, *bn_w, *bn_b, *fc_w, *fc_b, *f0_w, *f0_b, *f1_w, *f1_b;
    float *run_mean, *run_var; int64_t* nbt;
    float *taps, *ah, *al, *xhat, *rstd, *z, *att;
    int use_running;                         /* 1: inference-mode BatchNorm (module.eval()): normalise with run_mean / run_var, buffers untouched */
} TdrSfDynVecDesc;
int tdr_sf_dyn_vec_fwd(const TdrSfDynVecDesc* d, void* stream);
typedef struct TdrSfDynVecBwdDesc {
    int N, c, GK, KK, d;
    const float *ap, *wconv, *bn_w, *fc_w, *f0_w, *f1_w, *taps, *xhat, *rstd, *z, *att, *dtaps, *dah, *dal;
    float *dap, *g_wconv, *g_bn_w, *g_bn_b, *g_fc_w, *g_fc_b, *g_f0_w, *g_f0_b, *g_f1_w, *g_f1_b;
    float* ws;                               /* N * (2c + d + GK) floats */
} TdrSfDynVecBwdDesc;
int tdr_sf_dyn_vec_bwd(const TdrSfDynVecBwdDesc* d, void* stream);
/* low = k x k stencil of the reflection-padded planes with taps[n][c / (C / groups)]; mix = x * ah[n][c] + low * (al - ah)[n][c]
 * (= high * ah + low * al with high = x - low): the operand of SFconv's `out` convolution */
int tdr_sf_dynfilt_fwd(const float* x, int64_t x_ns, const float* taps, const float* ah, const float* al, int N, int C, int groups, int H,
                       int W, int k, float* low, float* mix, void* stream);
/* backward, reductions: dah[n][c] = sum dmix * (x - low), dal = sum dmix * low, dtaps[n][g][t] = sum_{c in g} (al - ah) sum_px dmix * xpad(t) */
int tdr_sf_dynfilt_bwd_reduce(const float* dmix, const float* x, int64_t x_ns, const float* low, const float* ah, const float* al, int N, int C,
                              int groups, int H, int W, int k, float* dah, float* dal, float* dtaps, void* stream);
/* backward, data: dx = dmix * ah + (al - ah) * stencil^T(dmix; taps) + dap[n][c] / (H W)   (dap: gradient of the pooled vector) */
int tdr_sf_dynfilt_bwd_dx(const float* dmix, const float* taps, const float* ah, const float* al, const float* dap, int N, int C, int groups,
                          int H, int W, int k, float* dx, int64_t dx_ns, void* stream);
/* ConvTranspose2d(Cin, Cout, 4, stride 2, padding 1) (BasicConv(transpose=True), :87) == 3x3 / pad 1 convolution to 4 Cout channels
 * (channel co * 4 + a * 2 + b = output parity (a, b)) + PixelShuffle(2): w [Cin][Cout][4][4] -> w3 [4 Cout][Cin][3][3], b -> b4 [4 Cout];
 * and the gradients back (dw3 -> dw, db4 -> db summed over the four parities). */
int tdr_convt4_weight_to_3x3(const float* w, const float* b, int Cin, int Cout, float* w3, float* b4, void* stream);
int tdr_convt4_grad_from_3x3(const float* dw3, const float* db4, int Cin, int Cout, float* dw, float* db, void* stream);
/* mode[0] == 'test' of the same network (inference with TLSC pooling, sfnet_arch_utils.py:108-113, :226-229, :247-250; forward only): the pooled
 * operand of Gap / Patch_ap / SFconv is the box-mean MAP of tdr_local_avgpool.
 * region_split: out [N][(c q + p1) q + p2][H / q][W / q] = the q x q region planes of a dense-NCHW view (q = 1: dense copy of a channel slice);
 * local_affine: y = m * pl[J] + (x - m) * (ph[J] + shift), m [N][C q q][H / q][W / q] the box-mean map of the region planes (Gap: q 1, shift 1,
 *   (fscale_h, fscale_d); Patch_ap: q 2, shift 0, (h, l));  emerge: out = low + (x - low) (SFconv, :218);
 * softmax_mix: per pixel softmax over the 2C logits [lh ; ll], mix = (x - low) * a_high + low * a_low (:226-232). */
int tdr_sf_region_split(const float* x, int64_t x_ns, int q, int N, int C, int H, int W, float* out, void* stream);
int tdr_sf_local_affine(const float* x, int64_t x_ns, const float* m, const float* ph, const float* pl, float shift, int q, int N, int C, int H, int W,
                        float* y, int64_t y_ns, void* stream);
int tdr_sf_emerge(const float* x, int64_t x_ns, const float* low, int N, int C, int HW, float* out, void* stream);
int tdr_sf_softmax_mix(const float* x, int64_t x_ns, const float* low, const float* lh, const float* ll, int N, int C, int HW, float* mix, void* stream);

#ifdef __cplusplus
}
#endif
#endif
