// STC connector (stock VideoLLaMA2 projector): the pieces that are not a GEMM or a row norm.
// /root/reference/streammind/model/multimodal_projector/builder.py:574-749 (STCConnector) builds
//   s1 = timm RegStage(depth 4, LayerNorm2d, SiLU) -> Conv3d(k = s = downsample, padding 1) + SiLU -> s2 = RegStage -> readout MLP
// and runs it on channel-first tensors.  Here every tensor is position-major ("NHWC": one row = one (frame, y, x) position, the
// channels contiguous), which is the layout the tower already produces and the layout the GEMMs want:
//   1x1 convolutions            -> sm_linear on the rows (bf16 operands, fp32 accumulation)
//   LayerNorm2d (+ SiLU)        -> sm_norm over the channels of a row (post_act SM_ACT_SILU)
//   depthwise 3x3 convolution   -> sm_dwconv3x3_nhwc (below): HBM-bound, 9 taps of fp32 per output
//   squeeze-excite              -> sm_pool_rows (mean over the positions of a frame) -> two small sm_linear -> sm_se_scale (below)
//   residual add + SiLU         -> sm_add_act (below)
//   Conv3d with stride = kernel -> sm_conv3d_patches (below: zero-padded gather of the kt x kh x kw neighbourhood of every output
//                                  position into one row) -> sm_linear with the weight flattened tap-major
// All of them stream their operands once; none is worth an MFMA.
#include "common.h"
#include "host.h"

// x fp32 [F][H][W][C], w fp32 [9][C] (tap-major: tap = (dy + 1) * 3 + (dx + 1)), stride 1, zero padding 1, no bias (timm ConvNormAct)
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, int H, int W, int C, const float* __restrict__ w,
                                                         float* __restrict__ out, size_t total4) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int c4 = C >> 2;
    const int c = (int)(t % c4) * 4;
    const size_t pos = t / c4;                       // (f * H + y) * W + x
    const int xq = (int)(pos % W), y = (int)((pos / W) % H);
    const size_t f = pos / ((size_t)W * H);
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = xq + dx;
            if (xx < 0 || xx >= W) continue;
            const f32x4 v = *(const f32x4*)(x + ((f * H + yy) * W + xx) * (size_t)C + c);
            const f32x4 k = *(const f32x4*)(w + (size_t)((dy + 1) * 3 + (dx + 1)) * C + c);
            acc[0] = fmaf(v[0], k[0], acc[0]); acc[1] = fmaf(v[1], k[1], acc[1]);
            acc[2] = fmaf(v[2], k[2], acc[2]); acc[3] = fmaf(v[3], k[3], acc[3]);
        }
    }
    *(f32x4*)(out + pos * (size_t)C + c) = acc;
}
extern "C" int sm_dwconv3x3_nhwc(const float* x, int F, int H, int W, int C, const float* w_tap_major, float* out, void* stream) {
    SM_REQUIRE(x && w_tap_major && out && F > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "sm_dwconv3x3_nhwc: bad args (C %% 4 == 0) F=%d H=%d W=%d C=%d", F, H, W, C);
    const size_t total4 = (size_t)F * H * W * (C >> 2);
    dwconv3x3_kernel<<<(unsigned)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, H, W, C, w_tap_major, out, total4);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// timm SEModule tail: out[r][c] = x[r][c] * sigmoid(gate[r / P][c]); 16-bit and / or fp32 output
__global__ __launch_bounds__(256) void se_scale_kernel(const float* __restrict__ x, const float* __restrict__ gate, int P, int C,
                                                        bf16_t* __restrict__ o16, float* __restrict__ o32, int f16, size_t total4) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int c4 = C >> 2;
    const int c = (int)(t % c4) * 4;
    const size_t r = t / c4;
    const f32x4 v = *(const f32x4*)(x + r * (size_t)C + c);
    const f32x4 g = *(const f32x4*)(gate + (r / P) * (size_t)C + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] * sigmoidf_(g[j]);
    if (o32) *(f32x4*)(o32 + r * (size_t)C + c) = o;
    if (o16) *(u32x2*)(o16 + r * (size_t)C + c) = u32x2{pack16_rt(o[0], o[1], f16), pack16_rt(o[2], o[3], f16)};
}
extern "C" int sm_se_scale(const float* x, const float* gate_logits, int F, int P, int C, void* out_16, float* out_f32, int op_dtype, void* stream) {
    SM_REQUIRE(x && gate_logits && (out_16 || out_f32) && F > 0 && P > 0 && C > 0 && (C & 3) == 0, "sm_se_scale: bad args");
    const size_t total4 = (size_t)F * P * (C >> 2);
    se_scale_kernel<<<(unsigned)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, gate_logits, P, C, (bf16_t*)out_16, out_f32, op_dtype == SM_OP_F16, total4);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// Bottleneck tail: out = act(a + b) (timm regnet Bottleneck.forward: x = x + shortcut; x = act3(x)); 16-bit and / or fp32 output
__global__ __launch_bounds__(256) void add_act_kernel(const float* __restrict__ a, const float* __restrict__ b, int act,
                                                       float* __restrict__ o32, bf16_t* __restrict__ o16, int f16, size_t n4) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    const f32x4 u = *(const f32x4*)(a + t * 4), v = b ? *(const f32x4*)(b + t * 4) : f32x4{0, 0, 0, 0};      // b == NULL: act(a), e.g. a plain 16-bit cast
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = apply_act(u[j] + v[j], act);
    if (o32) *(f32x4*)(o32 + t * 4) = o;
    if (o16) *(u32x2*)(o16 + t * 4) = u32x2{pack16_rt(o[0], o[1], f16), pack16_rt(o[2], o[3], f16)};
}
extern "C" int sm_add_act(const float* a, const float* b, size_t n, int act, float* out_f32, void* out_16, int op_dtype, void* stream) {
    SM_REQUIRE(a && (out_f32 || out_16) && n > 0 && (n & 3) == 0, "sm_add_act: bad args (n %% 4 == 0)");
    const size_t n4 = n >> 2;
    add_act_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(a, b, act, out_f32, (bf16_t*)out_16, op_dtype == SM_OP_F16, n4);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// nn.Conv3d(kernel = stride = (kt, kh, kw), padding = pad) as a GEMM (builder.py:608-617): 16-bit x [B][T][H][W][C] -> rows
// [B * To * Ho * Wo][kt * kh * kw * C], row = output position, column = tap-major (((dt * kh) + dy) * kw + dx) * C + c, zeros
// where the tap falls into the padding.  To = (T + 2 pad - kt) / kt + 1, likewise Ho, Wo.  One thread moves 8 channels (16 B).
__global__ __launch_bounds__(256) void conv3d_patches_kernel(const bf16_t* __restrict__ x, int T, int H, int W, int C, int kt, int kh, int kw,
                                                              int pad, int To, int Ho, int Wo, bf16_t* __restrict__ out, size_t total8) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total8) return;
    const int c8 = C >> 3;
    const int c = (int)(t % c8) * 8;
    size_t r = t / c8;
    const int taps = kt * kh * kw;
    const int tap = (int)(r % taps); r /= taps;
    const int dx = tap % kw, dy = (tap / kw) % kh, dt = tap / (kw * kh);
    const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho), to = (int)((r / ((size_t)Wo * Ho)) % To);
    const size_t b = r / ((size_t)Wo * Ho * To);
    const int ti = to * kt - pad + dt, yi = ho * kh - pad + dy, xi = wo * kw - pad + dx;
    u32x4 v = {0, 0, 0, 0};
    if (ti >= 0 && ti < T && yi >= 0 && yi < H && xi >= 0 && xi < W) v = *(const u32x4*)(x + ((((b * T + ti) * H + yi) * W + xi) * (size_t)C + c));
    *(u32x4*)(out + (r * taps + tap) * (size_t)C + c) = v;
}
extern "C" int sm_conv3d_patches(const void* x_16, int B, int T, int H, int W, int C, int kt, int kh, int kw, int pad, void* out_16, void* stream) {
    SM_REQUIRE(x_16 && out_16 && B > 0 && T > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && kt > 0 && kh > 0 && kw > 0 && pad >= 0 &&
               T + 2 * pad >= kt && H + 2 * pad >= kh && W + 2 * pad >= kw, "sm_conv3d_patches: bad args (C %% 8 == 0)");
    const int To = (T + 2 * pad - kt) / kt + 1, Ho = (H + 2 * pad - kh) / kh + 1, Wo = (W + 2 * pad - kw) / kw + 1;
    const size_t total8 = (size_t)B * To * Ho * Wo * kt * kh * kw * (C >> 3);
    conv3d_patches_kernel<<<(unsigned)((total8 + 255) / 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x_16, T, H, W, C, kt, kh, kw, pad, To, Ho, Wo,
                                                                                           (bf16_t*)out_16, total8);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// nn.AvgPool3d(kernel = stride = (kt, kh, kw)) + activation (STPConnector / SpatialPool sampler, builder.py:751-758,790-796):
// x fp32 [B][T][H][W][C] -> act(mean over each kt x kh x kw block) as fp32 and / or 16-bit [B][T/kt][H/kh][W/kw][C] (floor: a ragged
// border is dropped, as PyTorch does without ceil_mode)
__global__ __launch_bounds__(256) void avgpool3d_kernel(const float* __restrict__ x, int T, int H, int W, int C, int kt, int kh, int kw, int To, int Ho,
                                                         int Wo, int act, float* __restrict__ o32, bf16_t* __restrict__ o16, int f16, size_t total4) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int c4 = C >> 2;
    const int c = (int)(t % c4) * 4;
    const size_t r = t / c4;
    const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho), to = (int)((r / ((size_t)Wo * Ho)) % To);
    const size_t b = r / ((size_t)Wo * Ho * To);
    f32x4 acc = {0, 0, 0, 0};
    for (int dt = 0; dt < kt; ++dt)
        for (int dy = 0; dy < kh; ++dy)
            for (int dx = 0; dx < kw; ++dx) {
                const f32x4 v = *(const f32x4*)(x + ((((b * T + to * kt + dt) * H + ho * kh + dy) * W + wo * kw + dx) * (size_t)C + c));
                acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
            }
    const float inv = 1.0f / (float)(kt * kh * kw);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = apply_act(acc[j] * inv, act);
    if (o32) *(f32x4*)(o32 + r * (size_t)C + c) = o;
    if (o16) *(u32x2*)(o16 + r * (size_t)C + c) = u32x2{pack16_rt(o[0], o[1], f16), pack16_rt(o[2], o[3], f16)};
}
extern "C" int sm_avgpool3d_nhwc(const float* x, int B, int T, int H, int W, int C, int kt, int kh, int kw, int act, float* out_f32, void* out_16,
                                 int op_dtype, void* stream) {
    SM_REQUIRE(x && (out_f32 || out_16) && B > 0 && C > 0 && (C & 3) == 0 && kt > 0 && kh > 0 && kw > 0 && T >= kt && H >= kh && W >= kw,
               "sm_avgpool3d_nhwc: bad args (C %% 4 == 0, at least one full block per axis)");
    const int To = T / kt, Ho = H / kh, Wo = W / kw;
    const size_t total4 = (size_t)B * To * Ho * Wo * (C >> 2);
    avgpool3d_kernel<<<(unsigned)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, T, H, W, C, kt, kh, kw, To, Ho, Wo, act, out_f32, (bf16_t*)out_16,
                                                                                      op_dtype == SM_OP_F16, total4);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
