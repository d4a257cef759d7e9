/* libstreammind_hip.so -- C ABI of the MI355X (gfx950) StreamMind streaming hot path.
 *
 * The reference (xinding-sys/StreamMind) is a pure-Python nn.Module composition and has NO FFI/plugin
 * layer; the boundary its callers use is a set of Python call signatures.  This header is
 * the C ABI a native replacement sits behind; each entry point names the reference interface whose
 * arithmetic it replaces (file:line relative to /root/reference).  Python mirrors of those signatures
 * (streammind_amd/) bind it with ctypes -- see INTEGRATION.md for the stub a maintainer would add.
 *
 * Conventions: every pointer is a DEVICE pointer unless its name ends in `_host`; sizes are element
 * counts; `stream` is a hipStream_t passed as void*; all calls are asynchronous on `stream` and return
 * 0 on success or a negative SM_E* code (text via sm_last_error()).  No torch types, no hidden host syncs.
 * bf16 tensors are raw uint16 storage.  Linear weights are consumed in the "packed" fragment-major
 * layout produced by sm_pack_weight() (see streammind_amd/csrc/common.h).
 */
#ifndef STREAMMIND_HIP_H
#define STREAMMIND_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SM_OK 0
#define SM_EINVAL (-1)
#define SM_EHIP (-2)
#define SM_ESTATE (-3)

#define SM_ACT_NONE 0
#define SM_ACT_QUICK_GELU 1 /* HF CLIP MLP activation x*sigmoid(1.702x)                       */
#define SM_ACT_LEAKY_RELU 2 /* F.leaky_relu slope 0.01: PreNet/PostNet builder.py:168,178      */
#define SM_ACT_SOFTPLUS 3   /* F.softplus on dt: mamba_simple.py:238                           */
#define SM_ACT_SILU 4
#define SM_ACT_GELU 5       /* nn.GELU() exact (erf): build_mlp of the STC readout builder.py:566-571 */
#define SM_ACT_SWIGLU_DUAL 6 /* sm_linear only, 16-bit x, M > 16 (fewer rows: the dual weight-streaming kernel, w2): w = [gate rows | up rows] (N = 2F, F %% 128 == 0, the layout of the fused gate|up image),
                              * out_bf16[m][0..F) = 16-bit(silu(gate + bias[c]) * (up + bias[F + c])) -- MistralMLP's act_fn(gate_proj(x)) * up_proj(x) in the
                              * product's epilogue (the 256 x 256 kernel pairs gate / up fragments in one lane); elsewhere the product + sm_swiglu */

#define SM_X_BF16 0
#define SM_X_F32 1
#define SM_W_BF16 0
#define SM_W_FP8 1
#define SM_W_FP8_MFMA 2   /* the SM_W_FP8 image; calls with > 16 rows quantise the activations per row to e4m3 and run fp8 x fp8 MFMA */
#define SM_OP_BF16 0  /* 16-bit operands (packed weights, x, 16-bit outputs) are bfloat16 */
#define SM_OP_F16 1   /* ... IEEE half: tiled GEMM only (the ViT's optional fp16 mode = the reference demo's precision, builder.py:54) */
#define SM_TILE_AUTO 0
#define SM_TILE_128 128
#define SM_TILE_256 256
#define SM_TILE_256x128 256128
#define SM_TILE_256_ONE_TILE_PER_BLOCK 2561   /* the 256x256 kernel without its persistent (tile-walking) variant */

const char* sm_last_error(void);
int sm_abi_version(void);
/* number of device-visible GPUs' CUs etc. are not needed by callers; kept minimal on purpose. */

/* ------------------------------------------------------------------------------------------------
 * Weights.  Replaces: nn.Linear / nn.Conv2d(k=s=14) weight storage loaded by load_pretrained_model
 * (streammind/model/builder.py:30-210) and load_mm_projector (multimodal_projector/builder.py:66-85).
 * ---------------------------------------------------------------------------------------------- */
size_t sm_packed_elems(int N, int K); /* bf16 elements of the packed image (N->x16, K->x32 padded) */
int sm_pack_weight(const void* w_bf16, int N, int K, int ldw, void* out_packed, void* stream);
/* weight-only fp8 (OCP e4m3) for the weight-streaming path (BASELINE config 5, opt-in): per-row scale max|w|/448 into
 * scale_out[N], fp8 image of sm_packed_fp8_bytes(N,K) bytes.  No reference counterpart (the reference is fp16/bf16). */
size_t sm_packed_fp8_bytes(int N, int K);
int sm_quant_pack_weight_fp8(const void* w_bf16, int N, int K, int ldw, void* out_packed_fp8, float* scale_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Linear:  Y[M,N] = epilogue( X[M,K] . W[N,K]^T ).  Replaces every torch F.linear / cuBLAS GEMM+GEMV on
 * the path (SURVEY 2.3 K2,K3,K5-K8,K10,K11).  M <= 32 takes the weight-streaming "skinny" kernels (HBM-bound, MFMA
 * 16x16x32 with the weights as the A operand); larger M the LDS-tiled MFMA GEMM.  fp8 weights stream as fp8 for M <= 32
 * (16-bit activations: M <= 64; row scale on the fp32 sums); for more rows the call first expands them to a bf16 scratch image (row scale folded in) and
 * runs the bf16 kernels (SM_W_FP8_MFMA: fp8 x fp8 on the matrix pipe instead, activations quantised per row).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sm_linear_t {
    const void* w;        /* packed bf16 [N][K]                                                     */
    const void* w2;       /* optional second packed weight (same N,K): out = act(X.W^T) * (X.W2^T)  */
                          /*   (SwiGLU gate/up; skinny path only)                                   */
    int N, K;
    const void* x;        /* activations, row-major [M][ldx], bf16 or fp32 (x_dtype)                */
    int x_dtype;          /* SM_X_BF16 / SM_X_F32 (fp32 only on the skinny path, M <= 32)            */
    int precise;          /* fp32 x only: split x into bf16 hi+lo and issue two MFMAs (~fp32 acts)  */
    int M, ldx;
    const float* bias;    /* [N] or NULL                                                            */
    int act;              /* SM_ACT_* applied after bias                                            */
    const float* residual;/* fp32 [*][ldr] added after act, or NULL (may alias out_f32)             */
    int ldr;
    float* out_f32;       /* [M][ldo] or NULL                                                       */
    void* out_bf16;       /* [M][ldo_bf16] or NULL                                                  */
    int ldo, ldo_bf16;
    /* row remap (patch-embed writes patch p of frame b to token row b*(P+1)+1+p and adds pos-embed):
     * if remap_in > 0: out_row = (m / remap_in) * remap_out + remap_off + m % remap_in and, when
     * residual != NULL, residual row = remap_off + m % remap_in (broadcast over frames).            */
    int remap_in, remap_out, remap_off;
    /* V-transposed side output (ViT QKV): for n >= vt_n0 the value goes to
     * vt[((m / vt_S) * ((N - vt_n0) / vt_dh) + (n - vt_n0) / vt_dh) * vt_dh + (n - vt_n0) % vt_dh][m % vt_S]
     * (row length vt_ld) instead of out_bf16.                                                      */
    void* vt;
    int vt_n0, vt_S, vt_dh, vt_ld;
    /* weight storage: SM_W_BF16 (packed bf16) or SM_W_FP8 / SM_W_FP8_MFMA (sm_quant_pack_weight_fp8 image + per-row scales).
     * Up to 16 rows both fp8 kinds stream the weights once and expand them in registers (bf16 activations); SM_W_FP8 does
     * the same up to 32 rows, 64 rows of 16-bit activations (the LDS-shared weight-streaming kernel; row scale on the fp32 sums).  Above that
     * SM_W_FP8 expands the image to a bf16 scratch and runs the bf16 GEMM (weight-only fp8: bf16 activations); above 16 rows SM_W_FP8_MFMA
     * quantises every activation row to e4m3 (scale max|x|/448, the weights' rule) and multiplies on the fp8 matrix instruction
     * (v_mfma_scale_f32_16x16x128_f8f6f4, twice the bf16 rate): y = sx[m] sw[n] sum_k qx qw.  Needs K % 128 == 0, else as SM_W_FP8. */
    int w_dtype;
    const float* w_scale;
    const float* w2_scale;
    /* fused RMSNorm of the activations (MistralRMSNorm in front of q/k/v, gate/up and lm_head at decode time):
     * norm_gamma != NULL: x is the raw fp32 residual stream [M][ldx] and the product runs on
     * bf16(norm_gamma * (x * rsqrt(mean(x^2) + norm_eps))) -- what sm_norm would have written -- without the extra launch.
     * Weight-streaming path only (M <= 16, M*K <= 16384, bf16 or fp8 weights, x_dtype SM_X_F32, precise = 0). */
    const float* norm_gamma;
    float norm_eps;
    /* tile choice of the LDS-tiled GEMM (M > 32): 0 = automatic (256x256 once that grid fills >= 3/4 of the chip, else
     * 128x128), SM_TILE_128 / SM_TILE_256 / SM_TILE_256x128 force one kernel.  A tuning and test knob: results are the
     * same up to fp32 summation order. */
    int tile_hint;
    /* SM_OP_BF16 (default) or SM_OP_F16: w is then the packed image of fp16 values (sm_pack_weight is type-agnostic: it moves
     * 16-bit words), x and out_bf16 hold fp16.  Same MFMA rate, fp32 accumulation either way. */
    int op_dtype;
    /* LayerNorm of the FINISHED output row, behind the epilogue ("post-LN"): post_ln_gamma != NULL (with post_ln_beta) asks for
     * post_ln_out[m][0..N) = 16-bit(LN(out_f32[m][:]) * gamma + beta) (op_dtype), the operand of the next product, next to out_f32.
     * Needs out_f32 (and out_bf16 == NULL: the fused slab passes do not write it), remap_in == 0 and no vt.  Where the product runs as split-K slabs -- few tiles (one frame through the tower: a wave
     * per row, N == 1024) or 17..32 rows on the weight-streaming path (the connector / gate pass: a block per row, N %% 1024 == 0) --
     * the slab sum, bias, activation, residual and the norm of the row are ONE pass; everywhere else the call ends with the sm_norm_ex
     * launch the caller would have made (the same two-pass arithmetic either way). */
    const float* post_ln_gamma;
    const float* post_ln_beta;          /* NULL: RMSNorm (gamma * x * rsqrt(mean(x^2) + eps)) instead of LayerNorm                  */
    float post_ln_eps;
    void* post_ln_out;                  /* 16-bit output [M][post_ln_ldo] or NULL                                                    */
    int post_ln_ldo;
    float* post_ln_out_f32;             /* fp32 output [M][post_ln_ldo] or NULL (the connector / gate products take fp32 rows)       */
    int post_ln_act;                    /* SM_ACT_* applied to the normalised value (the connector's leaky_relu(norm_f(.)))          */
    /* activations with repeated column groups (the event gate's repeat_kv in front of o_proj, seq-len 1: builder.py:553-562):
     * x_rep > 1: column k of the [M][K] operand is read from x[m][(k / (x_rep * x_rep_dh)) * x_rep_dh + k % x_rep_dh] -- x holds
     * K / x_rep columns.  Weight-streaming path only (M <= 32, fp32 x); x_rep and x_rep_dh powers of two, x_rep_dh >= 8. */
    int x_rep, x_rep_dh;
    /* LayerNorm FOLDED into the two products around it (round 6; the ViT at >= 21 frames per lane, where every product runs on the
     * 256 x 256 tile kernels and the LayerNorm would otherwise be a launch of its own: 100 MB each, 46 per tower pass):
     *   LN(x) W^T + b  =  rstd[m] * ( (x * gamma) W^T  -  mu[m] * (W gamma) )  +  (W beta + b)
     * PRODUCER (fold_stats_out != NULL; a post-LN call: out_f32, post_ln_gamma, post_ln_out, N % 256 == 0): instead of the LayerNorm the call
     *   writes post_ln_out[m][n] = 16-bit(out_f32[m][n] * post_ln_gamma[n]) -- ONE rounding of the activation, as the LayerNorm's output has --
     *   and fold_stats_out[(m * (N / 256) + t) * 2 + {0, 1}] = (sum, sum of squares) of row m over the 256 columns of column tile t (fp32).
     * CONSUMER (fold_stats_in != NULL; 16-bit output only, N % 256 == 0, act none / quick_gelu): x holds those raw scaled rows of width K,
     *   fold_stats_in their K / 256 partial sums per row; mu / rstd come from them (tiles summed in order, E[x^2] - mu^2, + fold_eps), `bias`
     *   is ignored and the epilogue is  rstd * (acc - mu * fold_g[n]) + fold_c[n]  with fold_g = W gamma, fold_c = W beta + b (fp32 [N], the
     *   caller's, made once: sm_model_finalize does it for the tower).
     * Both sides exist on the 256 x 256 tile kernels only (>= 192 tiles, or tile_hint SM_TILE_256): anything else is SM_EINVAL, never a
     * silently LayerNorm-less product. */
    float* fold_stats_out;
    const float* fold_stats_in;
    const float* fold_g;
    const float* fold_c;
    float fold_eps;
} sm_linear_t;
int sm_linear(const sm_linear_t* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Vector / normalisation ops
 * ---------------------------------------------------------------------------------------------- */
/* LayerNorm (beta != NULL) or RMSNorm (beta == NULL) over rows of fp32 x[M][D]; optional leaky_relu on the
 * result; writes fp32 and/or bf16.  Replaces nn.LayerNorm (CLIP, mamba Block, norm_fn) and MistralRMSNorm. */
int sm_norm(const float* x, int M, int D, int ldx, const float* gamma, const float* beta, float eps,
            int post_act, float* out_f32, void* out_bf16, int ldo, void* stream);
/* the same with the 16-bit output written as SM_OP_F16 (IEEE half) or SM_OP_BF16 */
int sm_norm_ex(const float* x, int M, int D, int ldx, const float* gamma, const float* beta, float eps,
               int post_act, float* out_f32, void* out_bf16, int ldo, int op_dtype, void* stream);

/* a1 (mm_utils.py:449-464 + video_score_stream_demo.py:86): u8 HWC frames [B][H][W][3] -> bf16 patch
 * matrix [B*(H/p)*(W/p)][ldp], column c*p*p + i*p + j, value (u8/255 - mean[c]) / std[c]; columns
 * >= 3*p*p zero-filled up to ldp.  Optionally also the CHW fp32 pixel tensor (pixel_values).        */
int sm_preprocess_patches(const uint8_t* frames, int B, int H, int W, int patch, const float* mean3_host,
                          const float* std3_host, void* patches_bf16, int ldp, float* pixel_values_opt,
                          int op_dtype, void* stream);      /* op_dtype: 16-bit type of the patch matrix (SM_OP_BF16 / SM_OP_F16) */
/* same patch matrix from ALREADY-normalised pixel_values [B][3][H][W] (dtype SM_DT_BF16 / SM_DT_F32 / SM_DT_F16): the
 * reference-convention input of CLIPVisionTower.forward (clip_encoder.py:41-53; video_score_stream_demo.py:86) */
/* f2 ingest front-end (mm_utils.py:257-268 expand2square; 452-464 -> HF CLIPImageProcessor bicubic shortest-edge resize +
 * centre crop, i.e. PIL ImagingResample on 8-bit pixels): u8 frames [B][H][W][3] of ANY size -> u8 [B][out][out][3],
 * bit-exact with PIL.  pad_square: paste on a square canvas of pad_rgb (host, 3 bytes; the reference uses
 * int(image_mean * 255)) first.  tmp: caller-provided scratch of sm_ingest_tmp_bytes() bytes (the horizontal pass).
 * The first call for a new (size -> size) pair builds + uploads its coefficient tables synchronously.             */
size_t sm_ingest_tmp_bytes(int B, int H, int W, int pad_square, int out_size);
int sm_ingest_frames(const uint8_t* frames, int B, int H, int W, int pad_square, const uint8_t* pad_rgb_host, int out_size,
                     uint8_t* out_frames, uint8_t* tmp, void* stream);
int sm_patchify_pixels(const void* pixel_values, int dtype, int B, int H, int W, int patch, void* patches_bf16,
                       int ldp, int op_dtype, void* stream);
/* builder.py:405 on caller-held features: feats [T][P][C] (bf16/f32/f16) -> pooled fp32 [T][C] = mean over P */
int sm_pool_rows(const void* feats, int dtype, int T, int P, int C, float* pooled, void* stream);
/* CLS row: x[b*S + 0][:] = class_embedding + pos[0]  (HF CLIPVisionEmbeddings) */
int sm_vit_cls_rows(float* x, int B, int S, int D, const float* cls, const float* pos0, void* stream);
/* non-causal MHA over bf16 qkv [B*S][3*H*dh] (Q|K|V); ctx bf16 [B*S][H*dh].  V is taken row-major from qkv and
 * transposed while it is staged in LDS (vt == NULL), or read from a pre-transposed vt[B][H][dh][vt_ld].
 * Replaces HF CLIPAttention (eager / SDPA).                                                          */
int sm_vit_attention(const void* qkv, const void* vt, void* ctx, int B, int S, int H, int dh, int vt_ld, int op_dtype,
                     void* stream);     /* op_dtype SM_OP_F16: qkv / ctx hold fp16 (vt == NULL, dh == 64 only) */
/* builder.py:405 mean over the P patch tokens (CLS row skipped): x fp32 [B*S][D] -> pooled fp32 [B][D];
 * optionally the raw patch features as bf16 [B][P][D] (CLIPVisionTower.forward's return value).      */
int sm_pool_patches(const float* x, int B, int S, int D, float* pooled, void* feats_bf16_opt, void* stream);

/* Mamba recurrent step pieces (mamba_simple.py:208-253), M frames processed in order inside one launch:
 * conv: xz fp32 [M][2*di] (x = first di) , conv_state fp32 [di][d_conv] (rolled in place), w [di][d_conv],
 *       b [di] -> xc fp32 [M][di] = silu(conv)                                                        */
int sm_mamba_conv_step(const float* xz, int M, int di, int d_conv, float* conv_state, const float* conv_w,
                       const float* conv_b, float* xc, void* stream);
/* ssm:  h = exp(delta*A) h + delta*B*x ; y = h.C + D x ; y *= silu(z).  x_dbl fp32 [M][ldx] holds
 *       (dt_r | B | C) with B at column dt_rank; A = -exp(A_log).                                     */
int sm_mamba_ssm_step(const float* xc, const float* delta, const float* x_dbl, int ldx, int dt_rank,
                      const float* xz, int M, int di, int d_state, const float* A_log, const float* Dp,
                      float* ssm_state, float* y, void* stream);
/* STC connector pieces (builder.py:574-749, the stock VideoLLaMA2 projector; timm RegStage Bottleneck = 1x1 conv -> depthwise 3x3 ->
 * squeeze-excite -> 1x1 conv, LayerNorm2d + SiLU).  Position-major ("NHWC") rows: 1x1 convolutions are sm_linear, LayerNorm2d is
 * sm_norm over the channels of a row; these four are the rest (stc.hip).
 * depthwise 3x3, stride 1, zero padding 1, no bias: x fp32 [F][H][W][C], w fp32 [9][C] tap-major ((dy+1)*3 + dx+1) -> out fp32 */
int sm_dwconv3x3_nhwc(const float* x, int F, int H, int W, int C, const float* w_tap_major, float* out, void* stream);
/* SEModule tail: out[r][c] = x[r][c] * sigmoid(gate_logits[r / P][c]), rows r < F*P; 16-bit (op_dtype) and / or fp32 output */
int sm_se_scale(const float* x, const float* gate_logits, int F, int P, int C, void* out_16, float* out_f32, int op_dtype, void* stream);
/* Bottleneck tail: out = act(a + b) over n fp32 elements (n %% 4 == 0), SM_ACT_*; 16-bit (op_dtype) and / or fp32 output; b may be
 * NULL (out = act(a): with SM_ACT_NONE a plain fp32 -> 16-bit conversion) */
int sm_add_act(const float* a, const float* b, size_t n, int act, float* out_f32, void* out_16, int op_dtype, void* stream);
/* nn.Conv3d(kernel = stride = (kt,kh,kw), padding = pad) as a GEMM (builder.py:608-617): 16-bit x [B][T][H][W][C] -> 16-bit rows
 * [B*To*Ho*Wo][kt*kh*kw*C], column (((dt*kh)+dy)*kw+dx)*C + c, zeros in the padding; To = (T + 2 pad - kt)/kt + 1 etc. */
int sm_conv3d_patches(const void* x_16, int B, int T, int H, int W, int C, int kt, int kh, int kw, int pad, void* out_16, void* stream);
/* nn.AvgPool3d(kernel = stride = (kt,kh,kw)) + SM_ACT_* (STPConnector / SpatialPool sampler, builder.py:751-758,790-796): x fp32
 * [B][T][H][W][C] -> [B][T/kt][H/kh][W/kw][C] (floor), fp32 and / or 16-bit (op_dtype) */
int sm_avgpool3d_nhwc(const float* x, int B, int T, int H, int W, int C, int kt, int kh, int kw, int act, float* out_f32, void* out_16,
                      int op_dtype, void* stream);
/* repeat_kv for the seq-len-1 gate shortcut: v fp32 [M][KV*dh] -> out fp32 [M][H*dh], head h <- h/(H/KV) */
int sm_repeat_kv(const float* v, int M, int KV, int H, int dh, float* out, void* stream);
/* a9 (videollama2_arch.py:938-941): decision[m] = argmax(softmax(logits[m][0:2])), ties -> 0          */
int sm_gate_decide(const float* logits, int M, int32_t* decision, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LLM pieces (HF MistralForCausalLM.generate, videollama2_mistral.py:426-431)
 * ---------------------------------------------------------------------------------------------- */
/* a10 (videollama2_arch.py:951-984): out_f32[i] = table[ids[i]] if ids[i] >= 0 else tokens[-ids[i]-1]
 * (the host resolves the <video> sentinels into per-position frame indices encoded as negative ids).  */
int sm_embed_splice(const int32_t* ids, int n, const void* table_bf16, const float* tokens_f32, int D,
                    float* out_f32, void* stream);
/* RoPE (rotate_half convention) on q,k of qkv fp32 [n][(H+2KV)*dh] at positions pos0..pos0+n-1 using the
 * precomputed fp32 tables cos_tab/sin_tab [S_max][dh/2] (cos/sin(pos * theta^(-2j/dh)), built by the host
 * exactly as HF MistralRotaryEmbedding does); writes q bf16 [n][H*dh], appends k to kcache bf16
 * [S_max][KV*dh] and v to the transposed cache vtcache bf16 [KV][dh][S_max].                          */
int sm_rope_kv_append(const float* qkv, int n, int pos0, int H, int KV, int dh, const float* cos_tab,
                      const float* sin_tab, void* q_bf16, void* kcache, void* vtcache, int S_max, void* stream);
/* causal GQA attention of n new queries (positions pos0..) against the cache [0, pos0+n): ctx bf16 [n][H*dh] */
int sm_llm_attention(const void* q_bf16, const void* kcache, const void* vtcache, int n, int pos0, int H,
                     int KV, int dh, int S_max, void* ctx_bf16, void* stream);
/* Which kernel runs the causal prefill at head_dim 128 (process-wide; results are bit-identical either way): 1 = the round-6 prefill kernel (one 128-query
 * tile per block, longest tiles first, K / V^T tiles by LDS-DMA, fragment reads a batch ahead of the MFMAs), 0 = the general tile kernel, -1 = back to the
 * SM_ATTN_PREFILL environment variable / the default (1).  For A/B runs and the equality test. */
int sm_set_prefill_attention_kernel(int on);
/* single-token decode attention (flash-decoding: keys split over up to splits_max blocks per KV group, then merged);
 * q bf16 [H*dh] at position pos, cache as above; workspace fp32 [splits_max * H * (dh + 2)]                */
int sm_llm_attention_window(const void* q, const void* kcache, const void* vtcache, int n, int pos0, int H, int KV, int dh, int S_max, int window,
                            void* ctx, void* stream);       /* sm_llm_attention with Mistral's sliding window (keys (p - window, p]); window 0 = none */
int sm_llm_decode_attention_window(const void* q, const void* kcache, const void* vtcache, int pos, int H, int KV, int dh, int S_max, int window,
                                   float* workspace, int splits_max, void* ctx, void* stream);
int sm_llm_decode_attention(const void* q_bf16, const void* kcache, const void* vtcache, int pos, int H, int KV, int dh,
                            int S_max, float* workspace, int splits_max, void* ctx_bf16, void* stream);
/* out[m] = bf16( silu(gu[m][0:F]) * gu[m][F:2F] ) */
int sm_swiglu(const float* gu, int M, int F, void* out_bf16, void* stream);
/* greedy argmax over fp32 logits[V] -> token (int32, device) ; first max wins (torch.argmax)           */
int sm_argmax(const float* logits, int V, int32_t* out, void* stream);

/* ================================================================================================
 * Path-level API: model (weights + workspaces) and per-stream state.
 * Replaces, as one native object each:
 *   sm_model  <- what load_pretrained_model() returns as `model` (streammind/model/builder.py:30-210):
 *                CLIPVisionTower + Video_Mamba_seq (+ClsNet) + MistralForCausalLM weights.
 *   sm_stream <- the per-stream fields the reference keeps ON the model object
 *                (frame_feature, interval_id_list: language_model/videollama2_mistral.py:159-162) plus what
 *                the reference recomputes every frame: Mamba conv/ssm state, per-frame tokens, LLM KV cache.
 * One host thread per stream handle.  The sm_llm_* calls of a stream may be issued on ANOTHER HIP stream than its perception calls
 * (an "LLM lane": replies are decoded while the perception stream keeps consuming frames -- the two touch disjoint state: the
 * perception writes token-store rows >= the rows a prefill splices, the LLM its KV cache and activation buffers; sm_llm_prefill
 * orders its HIP stream behind the newest connector + gate pass by itself).  Different streams of one model may be driven concurrently on DIFFERENT HIP streams: all
 * per-call scratch is per sm_stream, and the vision tower's workspaces are kept per HIP stream (allocated on the first call a HIP
 * stream makes, never afterwards).  Two streams driven on two HIP streams fill each other's launch gaps and kernel tails
 * (measured +7 % aggregate frames/s at 28 frames per call each); calls issued on ONE HIP stream are ordered as usual.
 * A stream's token store (max_frames) and KV cache (max_seq) are fixed-size: a full store is an error, never a silent drop;
 * long-running deployments size them for the session (4096 tokens = 34 min at 2 fps, 0.5 GB of KV) or reset the stream
 * (sm_stream_reset; the reference's own callers reset by `model.frame_feature = None` between videos).
 * ============================================================================================== */
typedef struct sm_config_t {
    /* CLIP ViT (config.json of config.mm_vision_tower; clip_encoder.py:18-29) */
    int vit_image, vit_patch, vit_hidden, vit_heads, vit_mlp, vit_layers_run;
    float vit_eps;
    float img_mean[3], img_std[3];
    /* connector (multimodal_projector/builder.py:390-399, mamba_simple.py:31-58) */
    int conn_mm_hidden, conn_d_model, conn_d_state, conn_d_conv, conn_expand, conn_dt_rank;
    /* conn_d_state == 0 (with gate_layers == 0): a model WITHOUT the Mamba connector and the event gate -- tower + LLM only, for stock
     * VideoLLaMA2 checkpoints whose projector (the STC family) runs above this ABI and hands its tokens to sm_stream_write_tokens;
     * mm_projector.* tensors are then ignored by sm_model_load_tensor and the sm_stream_push_* / sm_group_push_* calls fail */
    float conn_eps;
    /* gate = ClsNet (builder.py:370-385) */
    int gate_hidden, gate_layers, gate_heads, gate_kv_heads, gate_mlp;
    float gate_eps;
    /* LLM (Mistral-7B config.json); llm_layers == 0 builds a perception-only model */
    int llm_hidden, llm_layers, llm_heads, llm_kv_heads, llm_mlp, llm_vocab;
    float llm_eps, llm_rope_theta;
    /* capacities */
    int max_frames_per_call; /* frames batched through sm_vit_encode in one call                       */
    int gate_precise;        /* 1: hi/lo bf16 activation split in the connector+gate GEMVs (~fp32 acts) */
    int weights_fp8;         /* 1: gate + LLM linear weights are quantised to fp8 (per-row scale) at load time and every  */
                             /*    decode/gate product streams them as fp8 (prefill chunks expand them to bf16 per call); */
                             /*    BASELINE config 5, opt-in: numerics differ from the bf16 checkpoint                     */
                             /* 2: the same images; calls with more than 16 rows (prefill chunks, teacher-forced evaluation) */
                             /*    quantise their activation rows to e4m3 and run fp8 x fp8 on the matrix pipe (SM_W_FP8_MFMA) */
    int vit_fp16;            /* 1: the vision tower's GEMM / attention operands (weights + activations) are IEEE fp16 instead */
                             /*    of bf16 -- the precision the reference's demo loads the model in (model/builder.py:54:      */
                             /*    torch_dtype=float16); same MFMA rate, fp32 accumulation and fp32 residual stream as before */
    int llm_fp16;            /* 1: the same for the LLM: linear weights, embedding table, activations, q / KV caches and the   */
                             /*    attention's P in IEEE fp16 (an fp16 checkpoint goes in bit for bit; bf16 storage would drop */
                             /*    3 of its mantissa bits and flip greedy token ids at near-ties).  Excludes weights_fp8.      */
    int proj_fp16;           /* 1: the same for the connector + event gate: their linear weights are kept as IEEE fp16 and the */
                             /*    fp32 activations enter the products as fp16 hi/lo pairs (gate_precise) -- an fp16 checkpoint */
                             /*    rounded to bf16 would move the gate logits by ~5e-3, five times the 1e-3 bar.  Excludes fp8. */
    int llm_sliding_window;  /* Mistral `sliding_window` (4096 for Mistral-7B-v0.1): a query at position p attends to keys        */
                             /*    (p - window, p] -- HF MistralModel's mask; 0 = full causal.  The KV cache stays linear (max_seq  */
                             /*    positions); prefill and decode attention read only the window.                                  */
} sm_config_t;

typedef struct sm_model sm_model;
typedef struct sm_stream sm_stream;

#define SM_DT_BF16 0
#define SM_DT_F32 1
#define SM_DT_F16 2

int sm_model_create(const sm_config_t* cfg, sm_model** out);
/* Hand one checkpoint tensor (DEVICE memory, row-major, HF/reference layout and name, e.g.
 * "model.mm_projector.mamba_model.ssms.0.mixer.in_proj.weight", "model.vision_tower.vision_tower.
 * vision_model.encoder.layers.3.mlp.fc1.weight", "model.layers.7.self_attn.q_proj.weight").
 * The library copies/packs it into its own storage; the caller may free `data` after the stream syncs.
 * Unknown names are rejected with SM_EINVAL unless they belong to a part the path never reads
 * (post_layernorm, vision layers >= vit_layers_run, gate q/k projections), which return 1 (= ignored). */
int sm_model_load_tensor(sm_model* m, const char* name, const void* data, int dtype, int ndim, const int64_t* shape,
                         void* stream);
/* checks that every tensor the path reads has been loaded; builds derived tables (RoPE) */
int sm_model_finalize(sm_model* m, void* stream);
void sm_model_destroy(sm_model* m);
/* weights_fp8 models: switch between weight-only fp8 (1) and fp8 x fp8 MFMA above 16 rows (2) at run time (same weight images) */
int sm_model_set_fp8_mode(sm_model* m, int mode);
/* names of tensors still missing, '\n'-separated, into buf (for error messages) */
int sm_model_missing(sm_model* m, char* buf, size_t buflen);

/* a1+a2(+K4): B frames u8 HWC [B][H][W][3] -> pooled fp32 [B][vit_hidden] (mean over patches of
 * hidden_states[-2], CLS dropped); optional raw patch features bf16 [B][P][vit_hidden]
 * (= CLIPVisionTower.forward, clip_encoder.py:41-53) and pixel_values fp32 [B][3][H][W].
 * Calls with more than 28 frames run the tower as lanes of at most one round of 256-row tiles (28 frames at 577 tokens on 256
 * CUs), TWO lanes at a time (the second on a side HIP stream of the caller's
 * stream, joined before the call's work on `stream` is complete from the caller's point of view; own workspaces): +4-9 %
 * frames/s, results bit-identical to separate calls of the lanes.  SM_VIT_LANES=1 in the environment keeps one lane.   */
int sm_vit_encode(sm_model* m, const uint8_t* frames, int B, float* pooled, void* feats_bf16_opt,
                  float* pixel_values_opt, void* stream);
/* LayerNorm folding of the tower (sm_linear_t.fold_*; lanes of >= 21 frames, where every product runs on the 256 x 256 tile kernels): process-wide switch.
 * -1 = the default -- the fp16 tower (vit_fp16) folds, the bf16 tower does not (its folded form sits 1.2e-3 from the matching-precision oracle, beyond the
 * 1e-3 asserted for the benchmarked dtype) --, 0 = never, 1 = both, -2 = back to the SM_VIT_LN_FOLD environment variable / the default. */
int sm_set_vit_ln_fold(int mode);
/* Frame lanes of a call with few frames (process-wide).  -1 = the default rule: a call whose out-proj / fc2 would be a little more than one or two whole rounds of
 * 128 x 128 tiles on this chip (8..10 and 15..20 frames of 577 tokens on 256 CUs) runs as TWO half batches on two HIP streams -- 4-9 % less time per call, results those
 * of the two half calls --, 1 = never, 2..8 = that many lanes for calls of up to SM_VIT_SMALL_MAX (8) frames, -2 = back to SM_VIT_SMALL_LANES / the default. */
int sm_set_vit_frame_lanes(int mode);
/* the same from normalised pixel_values [B][3][H][W] (what the reference's callers hand to CLIPVisionTower.forward) */
int sm_vit_encode_pixels(sm_model* m, const void* pixel_values, int dtype, int B, float* pooled, void* feats_bf16_opt,
                         void* stream);

int sm_stream_open(sm_model* m, int max_frames, int max_seq, sm_stream** out);
int sm_stream_reset(sm_stream* s, void* stream);
void sm_stream_close(sm_stream* s);
/* a5-a9 for M new frames given their pooled features: appends M per-frame tokens to the stream's token store,
 * writes gate logits fp32 [M][2] and decisions int32 [M] (device).  Recurrent form of the connector
 * (exactly the reference's full re-scan).                                            */
int sm_stream_push_pooled(sm_stream* s, const float* pooled, int M, float* logits, int32_t* decisions, void* stream);
/* convenience = sm_vit_encode + sm_stream_push_pooled (the per-frame "gate step") */
int sm_stream_push_frames(sm_stream* s, const uint8_t* frames, int M, float* logits, int32_t* decisions, void* stream);
/* pipelined form: the tower runs on `stream`, the connector + gate pass on the stream's own side HIP stream (behind the tower,
 * behind the previous call's pass), so `stream` can start the next call's tower at once and the memory-bound pass fills the
 * gaps of the next tower's MFMA kernels (~4 % more frames/s at 28 frames per call).  Same results.  logits / decisions are
 * complete for work enqueued on `stream` after sm_stream_join(s, stream); every other sm_stream_* / sm_llm_* / sm_group_* call
 * on the stream joins by itself. */
int sm_stream_push_frames_pipelined(sm_stream* s, const uint8_t* frames, int M, float* logits, int32_t* decisions, void* stream);
int sm_stream_join(sm_stream* s, void* stream);
/* per-call form for callers that keep several pipelined calls in flight (one batch of look-ahead): sm_stream_pass_ticket names the
 * pass of the LAST sm_stream_push_frames_pipelined (-1: none yet); sm_stream_join_ticket orders `stream` -- any HIP stream, e.g.
 * a read-back stream -- behind THAT pass only, not behind passes issued after it.  A ticket stays valid until two further
 * pipelined calls have been issued (then it names a newer pass: the wait is longer, never too short). */
int sm_stream_pass_ticket(sm_stream* s);
int sm_stream_join_ticket(sm_stream* s, int ticket, void* stream);
int sm_stream_num_frames(sm_stream* s);
const float* sm_stream_tokens(sm_stream* s);             /* device fp32 [num_frames][d_model]          */
int sm_stream_kv_len(sm_stream* s);
int sm_stream_set_kv_len(sm_stream* s, int n);            /* truncate the KV cache (prefix reuse)       */
/* tokens the stream's K / V cache holds room for RIGHT NOW.  The cache starts at min(max_seq, SM_KV_INITIAL_CAP = 512) tokens and is grown
 * (doubled, up to max_seq: reallocated and copied on the call's HIP stream) by the prefill / decode call that needs more -- a stream opened for
 * a 4096-token context costs 64 MB of cache until it fills it, so hundreds of open streams fit beside the model in 288 GB.  The prefill chunk
 * buffers are the MODEL's (one set per HIP stream a decoder call is issued on), not the stream's.  A call that has to grow the cache cannot be
 * captured into a hipGraph (it allocates): run the step eagerly once at that context first; and a graph captured BEFORE a growth holds the old
 * buffers' addresses -- re-capture after sm_stream_kv_capacity changes (a captured step bakes the host-side position as well, so it was never valid
 * beyond the context it was captured at). */
int sm_stream_kv_capacity(sm_stream* s);
/* a10+a12 prefill: n new positions; ids[i] >= 0 text token, ids[i] < 0 -> frame token (-ids[i]-1).
 * Appends to the KV cache at kv_len, leaves the greedy next token in the stream (device).           */
int sm_llm_prefill(sm_stream* s, const int32_t* ids_dev, int n, void* stream);
/* f1 teacher-forced forward (videollama2_mistral.py:173-259 with labels / llm_eval: super().forward over the spliced
 * sequence): the same n new positions as sm_llm_prefill, but the final norm + lm_head run on EVERY position;
 * logits_dev fp32 [n][vocab].  Appends to the KV cache like a prefill and leaves the greedy token of the last row. */
int sm_llm_forward_logits(sm_stream* s, const int32_t* ids_dev, int n, float* logits_dev, void* stream);
/* per-row softmax cross-entropy + argmax over logits fp32 [n][ld] (V valid columns): nll[i] = logsumexp(row i) -
 * row i[labels[i]], 0 where labels[i] == ignore_index; argmax = first maximal column (torch.argmax).  The caller passes
 * SHIFTED labels (row t scored against label t+1) and reduces nll as HF's CrossEntropyLoss does (mean over the scored
 * rows; class-weighted mean for the gate, builder.py:345-349).  Either output may be NULL.                   */
int sm_cross_entropy(const float* logits, int n, int V, int ld, const int32_t* labels, int ignore_index, float* nll,
                     int32_t* argmax, void* stream);
/* f1 "similarity" frame sampling (videollama2_arch.py:603-611: the top fraction of a clip's frame tokens by cosine similarity to the
 * LAST one): out[t] = cos(x[t], ref) for the T rows of x fp32 [T][ld >= D]; torch.nn.functional.cosine_similarity's arithmetic
 * (each norm clamped at 1e-8).  The caller ranks the T numbers.                                              */
int sm_cosine_rows(const float* x, int T, int D, int ld, const float* ref, float* out, void* stream);
/* a12 decode: n_steps greedy steps continuing from the last prefill/decode; out_ids_dev[n_steps] int32 device.
 * Step j emits the token predicted after the previous one, feeds it back, appends its KV.             */
int sm_llm_decode(sm_stream* s, int n_steps, int32_t* out_ids_dev, void* stream);
/* replace the pending token (the greedy argmax a prefill / decode step left) by the caller's choice, e.g. one SAMPLED from
 * sm_stream_logits (HF generate with do_sample=True: serve/model_worker.py:247-282 passes temperature / top_p); the next
 * sm_llm_decode step emits and feeds back THIS token */
int sm_stream_set_next_token(sm_stream* s, const int32_t* tok_dev, void* stream);
/* last-position logits of the most recent prefill/decode step: device fp32 [vocab] */
const float* sm_stream_logits(sm_stream* s);
/* async device-to-device copies out of the stream's own storage (tokens [t0, t0+n) x d_model fp32; vocabulary
 * logits fp32 [vocab] and the pending greedy token int32) */
int sm_stream_read_tokens(sm_stream* s, int t0, int n, float* out, void* stream);
int sm_stream_read_logits(sm_stream* s, float* out_opt, int32_t* next_token_out_opt, void* stream);
/* the connector's recurrent state after the frames pushed so far (Mamba.step's conv_state / ssm_state, mamba_simple.py:208-253):
 * conv fp32 [d_inner][d_conv] (the last d_conv inputs per channel, oldest first), ssm fp32 [d_inner][d_state]; either may be NULL.
 * For checkpoint / resume of a stream and for the long-horizon parity tests (the reference holds no such state: it re-scans
 * the whole history every frame, videollama2_arch.py:190-191). */
int sm_stream_read_state(sm_stream* s, float* conv_out, float* ssm_out, void* stream);
/* overwrite / append per-frame tokens [t0, t0+n) from caller-computed fp32 features (t0 <= num_frames): lets a
 * caller that already holds connector outputs (e.g. restored from a cache) seed the stream                 */
int sm_stream_write_tokens(sm_stream* s, int t0, int n, const float* src, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stream groups: one tick of S streams of ONE model as one ViT batch and one connector + gate weight pass.
 * Replaces S model objects each serving one frame per call (the reference holds the stream state on the model object and
 * supports batch size 1 only: language_model/videollama2_mistral.py:159-162, eval/video_score_stream_demo.py:283-299,
 * eval/inference_video_score_stream_ddp.py:325).  Per stream the results are those of its own sm_stream_push_frames calls
 * (same arithmetic; fp32 summation order of the skinny products may differ with the row count).  The group borrows the
 * streams: they stay usable on their own (LLM prefill / decode of a stream that fired), must outlive the group, and a
 * group call must not overlap another call on one of its streams.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sm_stream_group sm_stream_group;
int sm_group_create(sm_stream** streams, int S, sm_stream_group** out);
void sm_group_destroy(sm_stream_group* g);
int sm_group_size(sm_stream_group* g);
/* frames u8 [S][F][H][W][3] (the F new frames of stream 0, then of stream 1, ...), S*F <= max_frames_per_call, F <= 32 (16 with
 * fp8 weights); logits fp32 [S][F][2], decisions int32 [S][F] (either may be NULL) */
int sm_group_push_frames(sm_stream_group* g, const uint8_t* frames, int F, float* logits, int32_t* decisions, void* stream);
/* the same from pooled ViT features fp32 [S][F][vit_hidden] */
int sm_group_push_pooled(sm_stream_group* g, const float* pooled, int F, float* logits, int32_t* decisions, void* stream);
/* batched greedy decode: n_steps steps of every ACTIVE stream of the group (active_host[i] != 0; NULL = all; at most 512 active,
 * 16 with fp8 weights; equal max_seq) in one pass over the LLM weights per step (up to 32 streams: weight-streaming kernels, one row
 * per stream; 33..512: the linears as tiled MFMA GEMMs over all rows -- M = streams, where decode meets an MFMA roofline at all -- , RoPE + append
 * and attention in launches of 128 streams, token gather / arg-max 32 streams at a time).  Each active stream must hold a pending token
 * (its own sm_llm_prefill) and continues exactly as sm_llm_decode would: out_ids_dev int32 [S][n_steps] (rows of inactive
 * streams untouched), KV caches, positions, pending tokens and last logits advance per stream.  Replaces S batch-1 HF
 * generate loops (videollama2_mistral.py:426-431; "only support batch size 1"): batch-1 decode is bound by streaming 14.2 GB
 * of weights per token, which S streams share here. */
int sm_group_llm_decode(sm_stream_group* g, const int32_t* active_host, int n_steps, int32_t* out_ids_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * f2 ingest front-end, codec half: baseline JPEG (Motion-JPEG AVI frames, stills) -> RGB u8 in HBM (csrc/jpeg.hip).  Replaces the
 * decord / PIL decode in front of the streaming loop (eval/video_score_stream_demo.py:212-225, mm_utils.py:399-435) for the one codec
 * this image can decode.  The bit-serial Huffman stage runs on the host (one frame per call, thread-safe, no HIP call inside); the
 * dequantisation, the 8x8 inverse DCT, the chroma upsampling and the colour conversion run on the GPU with libjpeg's own integer
 * arithmetic (islow IDCT, fancy upsampling, 16-bit colour tables): byte for byte what PIL / libjpeg-turbo returns.
 * Baseline sequential, 8-bit, 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, restart intervals, implied (standard) Huffman tables;
 * anything else is SM_EINVAL with the reason in sm_last_error() and the caller keeps its host decoder.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sm_jpeg_info_t {
    int width, height, ncomp;
    int hs[3], vs[3];                 /* sampling factors per component (luma: 1 or 2; chroma 1)                              */
    int mcu_w, mcu_h, mcus_x, mcus_y;
    int blocks_x[3], blocks_y[3];     /* 8x8 blocks per component plane (padded to whole MCUs)                                */
    int coef_offset[3];               /* first coefficient of each component inside one frame's coefficient image            */
    int coef_count;                   /* int16 coefficients per frame                                                        */
} sm_jpeg_info_t;
int sm_jpeg_info(const uint8_t* data, size_t len, sm_jpeg_info_t* info);
/* host -> host: coefs int16 [coef_count] (natural order inside a block, blocks row-major per plane), qt uint16 [3][64] (natural order,
 * per component).  want != NULL: fail unless the frame has that geometry (frames of one clip share it). */
int sm_jpeg_decode_coefs(const uint8_t* data, size_t len, const sm_jpeg_info_t* want, int16_t* coefs, uint16_t* qt);
size_t sm_jpeg_planes_bytes(const sm_jpeg_info_t* info, int n_frames);
/* Entropy decode ON THE GPU for frames with restart intervals (round 5).  A restart interval (DRI: `restart` MCUs, RSTn markers between them) starts
 * byte-aligned with its DC predictors reset, i.e. it decodes independently of everything in front of it: one GPU lane per interval (a 720p 4:2:0 frame with
 * one interval per MCU row = 45 lanes; a batch of 28 frames = 1260), markers found by a parallel scan of the segment, the same ITU T.81 F.2.2 decode as
 * sm_jpeg_decode_coefs with the tables in LDS, on a copy of the interval's bytes with the stuffed zeros removed (one wave per interval).  Frames WITHOUT restart markers stay on the host path (sm_jpeg_scan_prepare says so).
 *   sm_jpeg_scan_prepare(data, len, want, scan)   host: markers only (no entropy decode) -> where the entropy-coded segment lies, DRI, the canonical Huffman
 *                                                 decode tables and the quantisation tables of the frame's ONE interleaved scan (or its one grey scan);
 *                                                 SM_EINVAL naming the reason for anything else (no DRI, per-component scans, progressive, ...)
 *   sm_jpeg_entropy_decode(bytes, bytes_total, offsets, scans, info, n, coefs, qt, status, stream)
 *                                                 device: bytes = the n files back to back (device copy, each file 16-byte aligned, 32 bytes of padding behind
 *                                                 the last), offsets[n] = each file's first byte,
 *                                                 scans[n] = the prepared scans (device copy) -> coefs int16 [n][coef_count] and qt uint16 [n][3][64] exactly
 *                                                 as sm_jpeg_decode_coefs leaves them; status int32 [n]: 0 or the first error of the frame (1 bad code,
 *                                                 2 run past the block, 3 missing / misnumbered RSTn, 4 segment ends early) -- the caller reads it after `stream`. */
typedef struct sm_jpeg_huff_t {       /* canonical decode table, device-friendly (T.81 F.2.2.3) */
    uint16_t fast[512];               /* 9-bit look-ahead: (length << 8) | symbol, 0 = longer code */
    int32_t maxcode[18];              /* largest code of each length (-1: none), [17] = sentinel */
    int32_t valoff[17];               /* valptr[l] - mincode[l] */
    uint8_t vals[256];
} sm_jpeg_huff_t;
typedef struct sm_jpeg_scan_t {
    uint32_t scan_offset, scan_len;   /* entropy-coded segment inside the file (bytes); scan_len runs to the end of the file */
    int32_t restart;                  /* MCUs per restart interval; 0 = no DRI (one serial stream: sm_jpeg_entropy_decode_sync) */
    int32_t n_intervals;              /* 0 when restart == 0 */
    int32_t ncomp;
    uint16_t qt[3][64];               /* per component, natural order */
    sm_jpeg_huff_t dc[3], ac[3];      /* per component */
} sm_jpeg_scan_t;
int sm_jpeg_scan_prepare(const uint8_t* data, size_t len, const sm_jpeg_info_t* want, sm_jpeg_scan_t* scan);
int sm_jpeg_entropy_decode(const uint8_t* bytes_dev, size_t bytes_total, const uint32_t* offsets_dev, const sm_jpeg_scan_t* scans_dev, const sm_jpeg_info_t* info, int n_frames,
                           int16_t* coefs_dev, uint16_t* qt_dev, int32_t* status_dev, void* stream);
/* The same for frames WITHOUT restart markers (restart == 0 in every frame's scan): the one serial Huffman stream of a frame is cut into subsequences of
 * 1024 bits, one GPU lane each.  A lane decodes its subsequence from a guessed decoder state and keeps a record (entry state, exit state, blocks completed);
 * in every round the lanes whose record does not enter where their predecessor's exits decode again from there and follow the stream downstream until they
 * reach an exit state already on record (a prefix code synchronises).  When the records chain (at most 16 rounds -- JS_ROUNDS in csrc/jpeg.hip; usually 1-2, status 5 beyond) a prefix sum numbers the
 * blocks, a last pass writes the coefficients and a scan turns the DC differences into values.  max_file_bytes: the longest file of the batch (bounds the
 * lanes per frame).  status as above, plus 5 = the records did not chain (the caller decodes that batch on the host). */
int sm_jpeg_entropy_decode_sync(const uint8_t* bytes_dev, size_t bytes_total, size_t max_file_bytes, const uint32_t* offsets_dev, const sm_jpeg_scan_t* scans_dev,
                                const sm_jpeg_info_t* info, int n_frames, int16_t* coefs_dev, uint16_t* qt_dev, int32_t* status_dev, void* stream);
/* diagnostic: rounds until the records of each frame of this stream's LAST sm_jpeg_entropy_decode_sync call chained (waits for the stream) */
int sm_jpeg_sync_rounds(void* stream, int32_t* rounds_host, int n_frames);
/* device: coefs [n][coef_count], qt [n][3][64] -> rgb u8 [n][height][width][3]; planes: scratch of sm_jpeg_planes_bytes bytes */
int sm_jpeg_reconstruct(const int16_t* coefs, const uint16_t* qt, const sm_jpeg_info_t* info, int n_frames, uint8_t* planes, uint8_t* rgb, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gated-token exchange between the GPUs of one node (csrc/comm.hip).  Replaces allgather_diff_shape
 * (/root/reference/streammind/dist.py:122-146: an all-gather of the per-rank row counts, a host read, then an all-gather of the rows
 * padded to the largest count) and its caller, the distributed evaluation loop (eval/inference_video_score_stream_ddp.py).
 * MI355X form: xGMI is point to point, so every rank writes its rows straight into a mailbox in every peer's HBM (hipIpc-mapped,
 * fine-grained), one hop, all links at once; a tick whose gate stayed silent moves one 16-byte header per peer and no payload.
 * One process per GPU; two processes on one GPU are a valid world (how the single-GPU test box exercises it).
 *
 *   sm_comm_init(rank, world, max_rows, row_bytes)     mailbox of 2 x world slots of max_rows x row_bytes (row_bytes % 16 == 0)
 *   sm_comm_export(handle_out[sm_comm_handle_bytes()])  this rank's mailbox handle; the CALLER moves the handles of all ranks
 *   sm_comm_connect(all_handles[world][handle_bytes])   (any transport, as an ncclUniqueId travels) and maps the peers here
 *   sm_comm_post(rows, n_rows, stream)                  tick t of this rank: n_rows (0 = silent) rows, device pointer, 16-B aligned
 *   sm_comm_collect(counts_out, payload_out, stream)    tick t of every rank: counts_out int32 [world] (device, may be NULL; -1 = that
 *                                                      rank did not arrive within SM_COMM_TIMEOUT_MS, default 5000), payload_out
 *                                                      [world][max_rows][row_bytes] (device, may be NULL): rows [0, count) of each rank
 *   sm_comm_host_counts(parity, counts_out[world])      the same counts from the pinned host mirror of tick parity (t & 1), valid once
 *                                                      `stream` has passed that collect; error if a rank timed out
 *   sm_comm_poll_counts(parity, counts_out, &missing)   the same read with a timeout as "not yet": 0 + counts, or 1 + the rank that had not
 *                                                      posted (no error recorded); then
 *   sm_comm_recollect(counts_out, payload_out, stream)  issues the collect of that same tick again -- ranks fire on different ticks and a
 *                                                      rank decoding a long reply lags by more than any GPU-side spin should last; the
 *                                                      caller owns the overall deadline (PeerWriteExchange: 30 min, the process-group default)
 *   sm_allgather_gated(...)                             post + collect of one tick (the blocking form)
 * post and collect alternate (post t, collect t, post t+1, ...); on one HIP stream per rank, or on streams ordered by events.
 * ---------------------------------------------------------------------------------------------- */
#define SM_COMM_MAX_RANKS 16
typedef struct sm_comm sm_comm;
int sm_comm_handle_bytes(void);
int sm_comm_init(int rank, int world, int max_rows, int row_bytes, sm_comm** out);
int sm_comm_export(sm_comm* c, void* handle_out);
int sm_comm_connect(sm_comm* c, const void* all_handles);
void sm_comm_destroy(sm_comm* c);
int sm_comm_post(sm_comm* c, const void* rows, int n_rows, void* stream);
int sm_comm_collect(sm_comm* c, int32_t* counts_out, void* payload_out, void* stream);
int sm_comm_host_counts(sm_comm* c, int tick_parity, int32_t* counts_out);
int sm_comm_poll_counts(sm_comm* c, int tick_parity, int32_t* counts_out, int* missing_rank);
int sm_comm_recollect(sm_comm* c, int32_t* counts_out, void* payload_out, void* stream);
int sm_allgather_gated(sm_comm* c, const void* rows, int n_rows, int32_t* counts_out, void* payload_out, void* stream);
int sm_comm_max_rows(sm_comm* c);
int sm_comm_tick(sm_comm* c);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py roofline leg; no reference counterpart -- the reference has only commented-out
 * time.time() pairs, builder.py:741-745).  Class bits: 0 tiled GEMM, 1 skinny linear, 2 attention.
 * While enabled, each launch of the class is bracketed by HIP events on its own stream.
 * ---------------------------------------------------------------------------------------------- */
int sm_prof_enable(int class_mask);
int sm_prof_reset(void);
int sm_prof_read(int cls, int* count, float* total_ms);   /* synchronises the recorded events */
/* the same over the launches recorded with one tag; tiled GEMM launches (class 0) carry ((long long)N << 32) | K of their product */
int sm_prof_read_tag(int cls, long long tag, int* count, float* total_ms);

#ifdef __cplusplus
}
#endif
#endif
