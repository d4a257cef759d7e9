// Small HBM/latency-bound ops of the path: normalisation, preprocessing, pooling, Mamba recurrent step,
// gate decision, embedding splice, RoPE + KV append, SwiGLU, argmax.
#include "common.h"
#include "host.h"

thread_local char g_sm_err[512] = {0};
extern "C" const char* sm_last_error(void) { return g_sm_err; }
extern "C" int sm_abi_version(void) { return 4; }       // 4: sm_linear_t grew by the fold_* fields (round 6)

// ------------------------------------------------------------------------------------------------ profiling hooks
#include <vector>
int g_sm_prof_mask = 0;
static std::vector<hipEvent_t> g_prof_ev[SM_PROF_NCLS];      // begin/end pairs in launch order
static std::vector<long long> g_prof_tag[SM_PROF_NCLS];      // one tag per pair (tiled GEMM launches: N << 32 | K)
static size_t g_prof_used[SM_PROF_NCLS] = {0, 0, 0};
static hipEvent_t prof_next(int cls) {
    if (g_prof_used[cls] == g_prof_ev[cls].size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        g_prof_ev[cls].push_back(e);
    }
    return g_prof_ev[cls][g_prof_used[cls]++];
}
void sm_prof_begin_(int cls, hipStream_t st, long long tag) {
    const size_t pair = g_prof_used[cls] >> 1;
    if (g_prof_tag[cls].size() <= pair) g_prof_tag[cls].resize(pair + 1);
    g_prof_tag[cls][pair] = tag;
    (void)hipEventRecord(prof_next(cls), st);
}
void sm_prof_end_(int cls, hipStream_t st) { (void)hipEventRecord(prof_next(cls), st); }
extern "C" int sm_prof_enable(int mask) { g_sm_prof_mask = mask; return SM_OK; }
extern "C" int sm_prof_reset(void) { for (int c = 0; c < SM_PROF_NCLS; ++c) g_prof_used[c] = 0; return SM_OK; }
static int prof_sum(int cls, bool by_tag, long long tag, int* count, float* total_ms) {
    SM_REQUIRE(cls >= 0 && cls < SM_PROF_NCLS && count && total_ms, "sm_prof_read: bad args");
    float tot = 0.f;
    int n = 0;
    for (size_t i = 0; i + 1 < g_prof_used[cls]; i += 2) {
        if (by_tag && g_prof_tag[cls][i >> 1] != tag) continue;
        SM_HIP(hipEventSynchronize(g_prof_ev[cls][i + 1]));
        float ms = 0.f;
        SM_HIP(hipEventElapsedTime(&ms, g_prof_ev[cls][i], g_prof_ev[cls][i + 1]));
        tot += ms; ++n;
    }
    *count = n; *total_ms = tot;
    return SM_OK;
}
extern "C" int sm_prof_read(int cls, int* count, float* total_ms) { return prof_sum(cls, false, 0, count, total_ms); }
extern "C" int sm_prof_read_tag(int cls, long long tag, int* count, float* total_ms) { return prof_sum(cls, true, tag, count, total_ms); }

// ------------------------------------------------------------------------------------------------ norm
// one wave per row; D <= 16384.  LayerNorm: two-pass (mean, then centred variance) from registers/L1.
template <bool LN>
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x, int M, int D, int ldx,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float eps, int post_act, float* __restrict__ of, bf16_t* __restrict__ ob,
                                                   int ldo, int f16) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    const int nv = D >> 2;     // D % 4 == 0
    float s = 0.f;
    for (int v = lane; v < nv; v += 64) {
        f32x4 t = *(const f32x4*)(xr + v * 4);
        s += LN ? (t[0] + t[1] + t[2] + t[3]) : (t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3]);
    }
    s = wave_sum(s);
    float mu = 0.f, rstd;
    if (LN) {
        mu = s / D;
        float q = 0.f;
        for (int v = lane; v < nv; v += 64) {
            f32x4 t = *(const f32x4*)(xr + v * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = t[j] - mu; q += d * d; }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / D + eps);
    } else {
        rstd = rsqrtf(s / D + eps);
    }
    for (int v = lane; v < nv; v += 64) {
        f32x4 t = *(const f32x4*)(xr + v * 4);
        f32x4 gm = *(const f32x4*)(gamma + v * 4);
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y;
            if (LN) y = (t[j] - mu) * rstd * gm[j] + beta[v * 4 + j];
            else y = gm[j] * (t[j] * rstd);
            o[j] = apply_act(y, post_act);
        }
        if (of) *(f32x4*)(of + (size_t)row * ldo + v * 4) = f32x4{o[0], o[1], o[2], o[3]};
        if (ob) *(u32x2*)(ob + (size_t)row * ldo + v * 4) = u32x2{pack16_rt(o[0], o[1], f16), pack16_rt(o[2], o[3], f16)};
    }
}

// many rows, D == 256 * NV: one wave per row, compile-time trip counts so the NV 16-byte loads of a lane are all in
// flight at once (the generic kernel's runtime-bounded loops serialise them: 88 % of wave cycles parked on loads)
template <bool LN, int NV>
__global__ __launch_bounds__(256) void norm_wave_fixed_kernel(const float* __restrict__ x, int M, int ldx,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, float* __restrict__ of, bf16_t* __restrict__ ob, int ldo, int f16) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx + lane * 4;
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = *(const f32x4*)(xr + j * 256);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        s += LN ? (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]) : (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
    s = wave_sum(s);
    float mu = 0.f, rstd;
    if (LN) {
        mu = s * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d = v[j][e] - mu; q += d * d; }
        q = wave_sum(q);
        rstd = rsqrtf(q * (1.0f / D) + eps);
    } else {
        rstd = rsqrtf(s * (1.0f / D) + eps);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + j * 256;
        f32x4 gm = *(const f32x4*)(gamma + c);
        f32x4 bt = {0, 0, 0, 0};
        if (LN) bt = *(const f32x4*)(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = LN ? (v[j][e] - mu) * rstd * gm[e] + bt[e] : gm[e] * (v[j][e] * rstd);
        if (of) *(f32x4*)(of + (size_t)row * ldo + c) = o;
        if (ob) *(u32x2*)(ob + (size_t)row * ldo + c) = u32x2{pack16_rt(o[0], o[1], f16), pack16_rt(o[2], o[3], f16)};
    }
}

// few rows (decode, gate step): one 256-thread block per row, the row lives in registers (D <= 8192), one HBM/L2 pass
template <bool LN>
__global__ __launch_bounds__(256) void norm_row_block_kernel(const float* __restrict__ x, int D, int ldx,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int post_act, float* __restrict__ of,
                                                             bf16_t* __restrict__ ob, int ldo, int f16) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* xr = x + (size_t)row * ldx;
    const int nv = D >> 2;
    f32x4 v[8], gm[8];
    float s = 0.f;
    // x and gamma loads all go out together: on the decode path this kernel is pure latency (one 16 KiB row), and fetching
    // gamma only after the reduction added a second L2 round trip
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = tid + j * 256;
        v[j] = f32x4{0, 0, 0, 0};
        gm[j] = f32x4{0, 0, 0, 0};
        if (c < nv) {
            v[j] = *(const f32x4*)(xr + c * 4);
            gm[j] = *(const f32x4*)(gamma + c * 4);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        s += LN ? (v[j][0] + v[j][1] + v[j][2] + v[j][3])
                : (v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2] + v[j][3] * v[j][3]);
    s = wave_sum(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3];
    float mu = 0.f, rstd;
    if (LN) {
        mu = s / D;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (tid + j * 256 < nv)
#pragma unroll
                for (int e = 0; e < 4; ++e) { float d = v[j][e] - mu; q += d * d; }
        q = wave_sum(q);
        if (lane == 0) red[4 + w] = q;
        __syncthreads();
        rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / D + eps);
    } else {
        rstd = rsqrtf(s / D + eps);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = tid + j * 256;
        if (c >= nv) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y = LN ? (v[j][e] - mu) * rstd * gm[j][e] + beta[c * 4 + e] : gm[j][e] * (v[j][e] * rstd);
            o[e] = apply_act(y, post_act);
        }
        if (of) *(f32x4*)(of + (size_t)row * ldo + c * 4) = f32x4{o[0], o[1], o[2], o[3]};
        if (ob) *(u32x2*)(ob + (size_t)row * ldo + c * 4) = u32x2{pack16_rt(o[0], o[1], f16), pack16_rt(o[2], o[3], f16)};
    }
}

extern "C" int sm_norm_ex(const float* x, int M, int D, int ldx, const float* gamma, const float* beta, float eps,
                          int post_act, float* out_f32, void* out_bf16, int ldo, int op_dtype, void* stream) {
    SM_REQUIRE(x && gamma && (out_f32 || out_bf16), "sm_norm: null arg");
    const int f16 = op_dtype == SM_OP_F16;
    SM_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, "sm_norm: D, ldx, ldo must be multiples of 4");
    hipStream_t st = (hipStream_t)stream;
    static int block_max_m = -1;       // SM_NORM_BLOCK_MAXM: largest M that takes one BLOCK per row (above it: one wave per row)
    if (block_max_m < 0) { const char* e = getenv("SM_NORM_BLOCK_MAXM"); block_max_m = e ? atoi(e) : 64; }
    if (M <= block_max_m && D <= 8192) {
        if (beta) norm_row_block_kernel<true><<<M, 256, 0, st>>>(x, D, ldx, gamma, beta, eps, post_act, out_f32, (bf16_t*)out_bf16, ldo, f16);
        else norm_row_block_kernel<false><<<M, 256, 0, st>>>(x, D, ldx, gamma, beta, eps, post_act, out_f32, (bf16_t*)out_bf16, ldo, f16);
        SM_LAUNCH_CHECK();
        return SM_OK;
    }
    if (post_act == 0 && (D == 1024 || D == 4096)) {
        const bf16_t* dummy = nullptr; (void)dummy;
        if (D == 1024) {
            if (beta) norm_wave_fixed_kernel<true, 4><<<cdiv(M, 4), 256, 0, st>>>(x, M, ldx, gamma, beta, eps, out_f32, (bf16_t*)out_bf16, ldo, f16);
            else norm_wave_fixed_kernel<false, 4><<<cdiv(M, 4), 256, 0, st>>>(x, M, ldx, gamma, beta, eps, out_f32, (bf16_t*)out_bf16, ldo, f16);
        } else {
            if (beta) norm_wave_fixed_kernel<true, 16><<<cdiv(M, 4), 256, 0, st>>>(x, M, ldx, gamma, beta, eps, out_f32, (bf16_t*)out_bf16, ldo, f16);
            else norm_wave_fixed_kernel<false, 16><<<cdiv(M, 4), 256, 0, st>>>(x, M, ldx, gamma, beta, eps, out_f32, (bf16_t*)out_bf16, ldo, f16);
        }
        SM_LAUNCH_CHECK();
        return SM_OK;
    }
    if (beta) norm_kernel<true><<<cdiv(M, 4), 256, 0, st>>>(x, M, D, ldx, gamma, beta, eps, post_act, out_f32, (bf16_t*)out_bf16, ldo, f16);
    else norm_kernel<false><<<cdiv(M, 4), 256, 0, st>>>(x, M, D, ldx, gamma, beta, eps, post_act, out_f32, (bf16_t*)out_bf16, ldo, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

extern "C" int sm_norm(const float* x, int M, int D, int ldx, const float* gamma, const float* beta, float eps,
                       int post_act, float* out_f32, void* out_bf16, int ldo, void* stream) {
    return sm_norm_ex(x, M, D, ldx, gamma, beta, eps, post_act, out_f32, out_bf16, ldo, SM_OP_BF16, stream);
}

// ------------------------------------------------------------------------------------------------ preprocess
struct Norm3 { float mean[3], istd[3]; };

// one thread per (patch row, 8 output columns).  u8 reads hit L2 (a frame is 338 KB); writes are 16 B/lane.
__global__ void preprocess_kernel(const uint8_t* __restrict__ fr, int B, int H, int W, int p, Norm3 nm,
                                  bf16_t* __restrict__ out, int ldp, float* __restrict__ pix, int f16) {
    const int gx = W / p, gy = H / p;
    const int cols8 = ldp >> 3;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)B * gx * gy * cols8;
    if (t >= total) return;
    int c8 = (int)(t % cols8);
    size_t prow = t / cols8;
    int b = (int)(prow / (gx * gy));
    int pi = (int)(prow % (gx * gy));
    int py = pi / gx, px = pi % gx;
    const int pp = p * p;
    uint32_t o[4];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int col = c8 * 8 + j;
        float val = 0.f;
        if (col < 3 * pp) {
            int c = col / pp, r = col % pp;
            int ii = r / p, jj = r % p;
            int y = py * p + ii, x = px * p + jj;
            float u = (float)fr[(((size_t)b * H + y) * W + x) * 3 + c];
            val = (u * (1.0f / 255.0f) - nm.mean[c]) * nm.istd[c];
            if (pix) pix[(((size_t)b * 3 + c) * H + y) * W + x] = val;
        }
        v[j] = val;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack16_rt(v[2 * j], v[2 * j + 1], f16);
    *(u32x4*)(out + prow * ldp + c8 * 8) = u32x4{o[0], o[1], o[2], o[3]};
}

extern "C" int sm_preprocess_patches(const uint8_t* frames, int B, int H, int W, int patch, const float* mean3,
                                     const float* std3, void* patches, int ldp, float* pix, int op_dtype, void* stream) {
    const int f16 = op_dtype == SM_OP_F16;
    SM_REQUIRE(frames && patches && mean3 && std3, "sm_preprocess_patches: null arg");
    SM_REQUIRE(B > 0 && patch > 0 && H % patch == 0 && W % patch == 0, "sm_preprocess_patches: H, W must be multiples of patch");
    SM_REQUIRE(ldp % 8 == 0 && ldp >= 3 * patch * patch, "sm_preprocess_patches: ldp must be a multiple of 8 and >= 3*p*p");
    Norm3 nm;
    for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3[c]; nm.istd[c] = 1.0f / std3[c]; }
    size_t total = (size_t)B * (H / patch) * (W / patch) * (ldp / 8);
    preprocess_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(frames, B, H, W, patch, nm,
                                                                                       (bf16_t*)patches, ldp, pix, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// pixel_values (already normalised, CHW, fp32 / bf16 / fp16) -> bf16 patch matrix: the reference-convention input of
// CLIPVisionTower.forward (clip_encoder.py:41-53), for callers that preprocess on the host like the reference does
template <int DT>
__device__ __forceinline__ float load_pix(const void* p, size_t i) {
    if (DT == 1) return ((const float*)p)[i];
    if (DT == 0) return bf2f(((const bf16_t*)p)[i]);
    return (float)((const _Float16*)p)[i];
}
template <int DT>
__global__ void patchify_pixels_kernel(const void* __restrict__ pix, int B, int H, int W, int p, bf16_t* __restrict__ out, int ldp, int f16) {
    const int gx = W / p, gy = H / p, cols8 = ldp >> 3, pp = p * p;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)B * gx * gy * cols8;
    if (t >= total) return;
    int c8 = (int)(t % cols8);
    size_t prow = t / cols8;
    int b = (int)(prow / (gx * gy)), pi = (int)(prow % (gx * gy));
    int py = pi / gx, px = pi % gx;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int col = c8 * 8 + j;
        float val = 0.f;
        if (col < 3 * pp) {
            int c = col / pp, r = col % pp;
            val = load_pix<DT>(pix, (((size_t)b * 3 + c) * H + py * p + r / p) * W + px * p + r % p);
        }
        v[j] = val;
    }
    *(u32x4*)(out + prow * ldp + c8 * 8) = u32x4{pack16_rt(v[0], v[1], f16), pack16_rt(v[2], v[3], f16), pack16_rt(v[4], v[5], f16), pack16_rt(v[6], v[7], f16)};
}
extern "C" int sm_patchify_pixels(const void* pix, int dtype, int B, int H, int W, int patch, void* patches, int ldp, int op_dtype,
                                  void* stream) {
    const int f16 = op_dtype == SM_OP_F16;
    SM_REQUIRE(pix && patches && B > 0 && patch > 0 && H % patch == 0 && W % patch == 0, "sm_patchify_pixels: bad args");
    SM_REQUIRE(ldp % 8 == 0 && ldp >= 3 * patch * patch && dtype >= 0 && dtype <= 2, "sm_patchify_pixels: ldp / dtype");
    size_t total = (size_t)B * (H / patch) * (W / patch) * (ldp / 8);
    unsigned blocks = (unsigned)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 1) patchify_pixels_kernel<1><<<blocks, 256, 0, st>>>(pix, B, H, W, patch, (bf16_t*)patches, ldp, f16);
    else if (dtype == 0) patchify_pixels_kernel<0><<<blocks, 256, 0, st>>>(pix, B, H, W, patch, (bf16_t*)patches, ldp, f16);
    else patchify_pixels_kernel<2><<<blocks, 256, 0, st>>>(pix, B, H, W, patch, (bf16_t*)patches, ldp, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// mean over the P rows of each of T groups: feats [T][P][C] (bf16 / fp32 / fp16) -> pooled fp32 [T][C]   (builder.py:405)
template <int DT>
__global__ __launch_bounds__(256) void pool_rows_kernel(const void* __restrict__ f, int P, int C, float* __restrict__ pooled) {
    __shared__ float red[4][64];
    const int t = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int r = w; r < P; r += 4) s += load_pix<DT>(f, ((size_t)t * P + r) * C + c);
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C) pooled[(size_t)t * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)P;
}
extern "C" int sm_pool_rows(const void* feats, int dtype, int T, int P, int C, float* pooled, void* stream) {
    SM_REQUIRE(feats && pooled && T > 0 && P > 0 && C > 0 && dtype >= 0 && dtype <= 2, "sm_pool_rows: bad args");
    dim3 grid(cdiv(C, 64), T);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 1) pool_rows_kernel<1><<<grid, 256, 0, st>>>(feats, P, C, pooled);
    else if (dtype == 0) pool_rows_kernel<0><<<grid, 256, 0, st>>>(feats, P, C, pooled);
    else pool_rows_kernel<2><<<grid, 256, 0, st>>>(feats, P, C, pooled);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

__global__ void cls_rows_kernel(float* x, int B, int S, int D, const float* cls, const float* pos0) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    int b = t / D, d = t % D;
    x[(size_t)b * S * D + d] = cls[d] + pos0[d];
}
extern "C" int sm_vit_cls_rows(float* x, int B, int S, int D, const float* cls, const float* pos0, void* stream) {
    SM_REQUIRE(x && cls && pos0 && B > 0, "sm_vit_cls_rows: bad args");
    cls_rows_kernel<<<cdiv(B * D, 256), 256, 0, (hipStream_t)stream>>>(x, B, S, D, cls, pos0);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// ------------------------------------------------------------------------------------------------ pooling
// block = (frame b, 64-column slab); 16 waves split the P patch rows (4 rows in flight per wave: one frame alone is a
// latency-bound 16-block launch), lanes own one column each -> coalesced 256 B rows
__global__ __launch_bounds__(1024) void pool_kernel(const float* __restrict__ x, int S, int D, float* __restrict__ pooled,
                                                    bf16_t* __restrict__ feats) {
    __shared__ float red[16][64];
    const int b = blockIdx.y, l = threadIdx.x & 63, c = blockIdx.x * 64 + l, w = threadIdx.x >> 6;
    const int P = S - 1;
    float s = 0.f;
    if (c < D) {
        const float* xb = x + ((size_t)b * S + 1) * D + c;
        for (int r0 = w; r0 < P; r0 += 64) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = r0 + u * 16 < P ? xb[(size_t)(r0 + u * 16) * D] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += v[u];
                if (feats && r0 + u * 16 < P) feats[((size_t)b * P + r0 + u * 16) * D + c] = (bf16_t)f2bf(v[u]);
            }
        }
    }
    red[w][l] = s;
    __syncthreads();
    if (w == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][l];
        pooled[(size_t)b * D + c] = t / (float)P;
    }
}
extern "C" int sm_pool_patches(const float* x, int B, int S, int D, float* pooled, void* feats, void* stream) {
    SM_REQUIRE(x && pooled && B > 0 && S > 1, "sm_pool_patches: bad args");
    pool_kernel<<<dim3(cdiv(D, 64), B), 1024, 0, (hipStream_t)stream>>>(x, S, D, pooled, (bf16_t*)feats);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// the same mean over the patch rows (CLS dropped) of a 16-bit matrix [B * S][D] -> fp32 [B][D]: the operand of the LAST tower layer's fc2
// when only the pooled feature is wanted -- mean_p(fc2(h_p)) = fc2(mean_p h_p), SURVEY 7 step 3.  Block = (frame, 128-column slab), a
// lane owns two columns (one 32-bit load per row), 16 waves split the rows, 8 rows in flight per wave.
__global__ __launch_bounds__(1024) void pool16_kernel(const bf16_t* __restrict__ h, int S, int D, float* __restrict__ out, int f16) {
    __shared__ float red[16][128];
    const int b = blockIdx.y, l = threadIdx.x & 63, c = blockIdx.x * 128 + l * 2, w = threadIdx.x >> 6;
    const int P = S - 1;
    float s0 = 0.f, s1 = 0.f;
    if (c < D) {
        const bf16_t* hb = h + ((size_t)b * S + 1) * D + c;
        for (int r0 = w; r0 < P; r0 += 128) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = r0 + u * 16 < P ? *(const uint32_t*)(hb + (size_t)(r0 + u * 16) * D) : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (f16) { s0 += (float)__builtin_bit_cast(_Float16, (uint16_t)(v[u] & 0xffffu)); s1 += (float)__builtin_bit_cast(_Float16, (uint16_t)(v[u] >> 16)); }
                else { s0 += __uint_as_float(v[u] << 16); s1 += __uint_as_float(v[u] & 0xffff0000u); }
            }
        }
    }
    red[w][2 * l] = s0; red[w][2 * l + 1] = s1;
    __syncthreads();
    if (threadIdx.x < 128 && blockIdx.x * 128 + threadIdx.x < D) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][threadIdx.x];
        out[(size_t)b * D + blockIdx.x * 128 + threadIdx.x] = t / (float)P;
    }
}
int sm_pool_patches16(const void* h, int B, int S, int D, float* out, int f16, void* stream) {
    SM_REQUIRE(h && out && B > 0 && S > 1 && (D & 1) == 0, "sm_pool_patches16: bad args");
    pool16_kernel<<<dim3(cdiv(D, 128), B), 1024, 0, (hipStream_t)stream>>>((const bf16_t*)h, S, D, out, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// ------------------------------------------------------------------------------------------------ mamba step
// rows are S segments (streams) of F consecutive frames each: blockIdx.y = segment, its recurrent state st.p[segment]
// DC > 0: d_conv is a compile-time constant (the window and the taps live in registers); DC = 0: run-time width (<= 8) -- with run-time
// trip counts the two arrays are indexed dynamically and live in scratch memory, which made this kernel 27 us per 28-frame pass
template <int DC>
__global__ void mamba_conv_kernel(const float* __restrict__ xz, int F, int di, int dc_rt, SmSegStates st,
                                  const float* __restrict__ cw, const float* __restrict__ cb, float* __restrict__ xc) {
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= di) return;
    const int dc = DC > 0 ? DC : dc_rt;
    float* __restrict__ cs = st.p[blockIdx.y];
    const int m0 = blockIdx.y * F;
    float stt[DC > 0 ? DC : 8], w[DC > 0 ? DC : 8];
#pragma unroll
    for (int j = 0; j < dc; ++j) { stt[j] = cs[(size_t)d * dc + j]; w[j] = cw[(size_t)d * dc + j]; }
    const float bias = cb[d];
    // the recurrence over a segment's frames is serial, its loads are not: eight frames' inputs are requested together (a load per
    // step on the critical path made 28 frames cost 28 memory round trips: 32 us per pass)
    for (int mb = m0; mb < m0 + F; mb += 8) {
        float xin[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xin[u] = mb + u < m0 + F ? xz[(size_t)(mb + u) * 2 * di + d] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (mb + u >= m0 + F) break;
#pragma unroll
            for (int j = 0; j + 1 < dc; ++j) stt[j] = stt[j + 1];          // torch.roll(shifts=-1); state[..., -1] = x
            stt[dc - 1] = xin[u];
            // the operation sequence is written out (no contraction left to the optimiser): a frame must get the same bits whether it is the
            // first or the fifth of its call -- the unrolled copies of this body were contracted differently
            float a = 0.f;
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int j = 0; j < dc; ++j) a = __builtin_fmaf(stt[j], w[j], a);
                a = a + bias;
            }
            xc[(size_t)(mb + u) * di + d] = siluf_(a);
        }
    }
#pragma unroll
    for (int j = 0; j < dc; ++j) cs[(size_t)d * dc + j] = stt[j];
}
int sm_mamba_conv_step_seg(const float* xz, int S, int F, int di, int d_conv, const SmSegStates& st, const float* conv_w,
                           const float* conv_b, float* xc, void* stream) {
    SM_REQUIRE(xz && conv_w && conv_b && xc && S > 0 && S <= SM_MAX_SEG && F > 0 && d_conv <= 8, "sm_mamba_conv_step: bad args");
    if (d_conv == 4) mamba_conv_kernel<4><<<dim3(cdiv(di, 256), S), 256, 0, (hipStream_t)stream>>>(xz, F, di, d_conv, st, conv_w, conv_b, xc);
    else mamba_conv_kernel<0><<<dim3(cdiv(di, 256), S), 256, 0, (hipStream_t)stream>>>(xz, F, di, d_conv, st, conv_w, conv_b, xc);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
extern "C" int sm_mamba_conv_step(const float* xz, int M, int di, int d_conv, float* conv_state, const float* conv_w,
                                  const float* conv_b, float* xc, void* stream) {
    SM_REQUIRE(conv_state, "sm_mamba_conv_step: null state");
    SmSegStates st;
    st.p[0] = conv_state;
    return sm_mamba_conv_step_seg(xz, 1, M, di, d_conv, st, conv_w, conv_b, xc, stream);
}

// thread per channel d, d_state (<= 32) state elements in registers; segments as in mamba_conv_kernel
// DS > 0: compile-time d_state (state and decay rates in registers); DS = 0: run-time (<= 32; the arrays then live in scratch, see the conv kernel)
template <int DS>
__global__ void mamba_ssm_kernel(const float* __restrict__ xc, const float* __restrict__ delta,
                                 const float* __restrict__ xdbl, int ldx, int R, const float* __restrict__ xz, int F, int di,
                                 int ds_rt, const float* __restrict__ Alog, const float* __restrict__ Dp,
                                 SmSegStates st, float* __restrict__ y) {
    // B and C of every frame of the segment (2 * d_state floats per frame, the same for all channels) go to LDS once: fetched per step
    // inside the serial loop they were a memory round trip per frame on the critical path (22 us of a 28-frame pass)
    __shared__ float bc[SM_MAX_SEG * 64];
    const int ds = DS > 0 ? DS : ds_rt;
    const int m0 = blockIdx.y * F;
    const bool staged = F <= SM_MAX_SEG;
    if (staged) {
        for (int t = threadIdx.x; t < F * 2 * ds; t += blockDim.x) {
            const int f = t / (2 * ds), j = t - f * 2 * ds;
            bc[f * 64 + j] = xdbl[(size_t)(m0 + f) * ldx + R + j];
        }
        __syncthreads();
    }
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= di) return;
    float* __restrict__ hst = st.p[blockIdx.y];
    float h[DS > 0 ? DS : 32], A[DS > 0 ? DS : 32];
#pragma unroll
    for (int n = 0; n < ds; ++n) { h[n] = hst[(size_t)d * ds + n]; A[n] = -__expf(Alog[(size_t)d * ds + n]); }
    const float Dd = Dp[d];
    // serial recurrence, batched loads (see mamba_conv_kernel): four frames' dt / x / z are in flight together (60 us -> per pass of 28 frames
    // was 28 dependent round trips)
    for (int mb = m0; mb < m0 + F; mb += 4) {
        float dtv[4], xvv[4], zv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = mb + u < m0 + F;
            dtv[u] = ok ? delta[(size_t)(mb + u) * di + d] : 0.f;
            xvv[u] = ok ? xc[(size_t)(mb + u) * di + d] : 0.f;
            zv[u] = ok ? xz[(size_t)(mb + u) * 2 * di + di + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (mb + u >= m0 + F) break;
            const int m = mb + u;
            const float dt = dtv[u], xv = xvv[u];
            float Bv[DS > 0 ? DS : 32], Cv[DS > 0 ? DS : 32];
            if (staged) {          // LDS (ds_read, not a generic pointer)
#pragma unroll
                for (int n = 0; n < ds; ++n) { Bv[n] = bc[(m - m0) * 64 + n]; Cv[n] = bc[(m - m0) * 64 + ds + n]; }
            } else {
#pragma unroll
                for (int n = 0; n < ds; ++n) { Bv[n] = xdbl[(size_t)m * ldx + R + n]; Cv[n] = xdbl[(size_t)m * ldx + R + ds + n]; }
            }
            float acc = 0.f, yv;
            {   // pinned operation sequence (see mamba_conv_kernel): the same bits for a frame wherever it sits in its call
#pragma clang fp contract(off)
                const float dtx = dt * xv;
#pragma unroll
                for (int n = 0; n < ds; ++n) {
                    h[n] = __builtin_fmaf(__expf(dt * A[n]), h[n], dtx * Bv[n]);
                    acc = __builtin_fmaf(h[n], Cv[n], acc);
                }
                yv = __builtin_fmaf(Dd, xv, acc);
            }
            y[(size_t)m * di + d] = yv * siluf_(zv[u]);
        }
    }
#pragma unroll
    for (int n = 0; n < ds; ++n) hst[(size_t)d * ds + n] = h[n];
}
int sm_mamba_ssm_step_seg(const float* xc, const float* delta, const float* x_dbl, int ldx, int dt_rank, const float* xz, int S,
                          int F, int di, int d_state, const float* A_log, const float* Dp, const SmSegStates& st, float* y,
                          void* stream) {
    SM_REQUIRE(xc && delta && x_dbl && xz && A_log && Dp && y, "sm_mamba_ssm_step: null arg");
    SM_REQUIRE(S > 0 && S <= SM_MAX_SEG && F > 0 && d_state <= 32, "sm_mamba_ssm_step: d_state <= 32, segments <= %d", SM_MAX_SEG);
    if (d_state == 16) mamba_ssm_kernel<16><<<dim3(cdiv(di, 128), S), 128, 0, (hipStream_t)stream>>>(xc, delta, x_dbl, ldx, dt_rank, xz, F, di, d_state, A_log, Dp, st, y);
    else mamba_ssm_kernel<0><<<dim3(cdiv(di, 128), S), 128, 0, (hipStream_t)stream>>>(xc, delta, x_dbl, ldx, dt_rank, xz, F, di, d_state, A_log, Dp, st, y);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
extern "C" int sm_mamba_ssm_step(const float* xc, const float* delta, const float* x_dbl, int ldx, int dt_rank,
                                 const float* xz, int M, int di, int d_state, const float* A_log, const float* Dp,
                                 float* ssm_state, float* y, void* stream) {
    SM_REQUIRE(ssm_state, "sm_mamba_ssm_step: null state");
    SmSegStates st;
    st.p[0] = ssm_state;
    return sm_mamba_ssm_step_seg(xc, delta, x_dbl, ldx, dt_rank, xz, 1, M, di, d_state, A_log, Dp, st, y, stream);
}

// rows [S][F][d] -> per-segment destinations dst.p[s] + f*d  (per-stream token stores)
__global__ void scatter_rows_kernel(const float* __restrict__ src, int F, int d, SmSegStates dst) {
    const int row = blockIdx.x, s = row / F, f = row - s * F;
    float* __restrict__ o = dst.p[s] + (size_t)f * d;
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) *(f32x4*)(o + c) = *(const f32x4*)(src + (size_t)row * d + c);
}
int sm_scatter_rows(const float* src, int S, int F, int d, const SmSegStates& dst, void* stream) {
    SM_REQUIRE(src && S > 0 && S <= SM_MAX_SEG && F > 0 && (d & 3) == 0, "sm_scatter_rows: bad args");
    scatter_rows_kernel<<<S * F, 256, 0, (hipStream_t)stream>>>(src, F, d, dst);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

__global__ void repeat_kv_kernel(const float* v, int M, int KV, int H, int dh, float* out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int tot = M * H * dh;
    if (t >= tot) return;
    int j = t % dh, h = (t / dh) % H, m = t / (dh * H);
    out[t] = v[((size_t)m * KV + h / (H / KV)) * dh + j];
}
extern "C" int sm_repeat_kv(const float* v, int M, int KV, int H, int dh, float* out, void* stream) {
    SM_REQUIRE(v && out && M > 0 && H % KV == 0, "sm_repeat_kv: bad args");
    repeat_kv_kernel<<<cdiv(M * H * dh, 256), 256, 0, (hipStream_t)stream>>>(v, M, KV, H, dh, out);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// Tail of the event gate in ONE launch (builder.py:553-562: final RMSNorm of the last hidden state, the 2-way `score` head, then
// softmax + argmax -- monotone, so a comparison): a block per row; the row lives in registers (D <= 8192), the head weights are the
// checkpoint's 16-bit values widened to fp32 (head_w [2][D], unpacked once at finalize), everything is fp32 FMA.  Replaces a norm
// launch, a 2-column weight-streaming product (17 us at 28 rows: one block walks all of K) and the decision launch.
__global__ __launch_bounds__(256) void gate_tail_kernel(const float* __restrict__ x, int D, int ldx, const float* __restrict__ gamma, float eps,
                                                        const float* __restrict__ head_w, float* __restrict__ logits, int32_t* __restrict__ dec) {
    __shared__ float red[3][4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* xr = x + (size_t)row * ldx;
    const int nv = D >> 2;
    f32x4 v[8], gm[8], w0[8], w1[8];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = tid + j * 256;
        v[j] = gm[j] = w0[j] = w1[j] = f32x4{0, 0, 0, 0};
        if (c < nv) {
            v[j] = *(const f32x4*)(xr + (size_t)c * 4);
            gm[j] = *(const f32x4*)(gamma + (size_t)c * 4);
            w0[j] = *(const f32x4*)(head_w + (size_t)c * 4);
            w1[j] = *(const f32x4*)(head_w + D + (size_t)c * 4);
        }
        q += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
    }
    q = wave_sum(q);
    if (lane == 0) red[0][w] = q;
    __syncthreads();
    const float rstd = rsqrtf(((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)D + eps);
    float l0 = 0.f, l1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float hn = gm[j][e] * (v[j][e] * rstd);
            l0 += hn * w0[j][e];
            l1 += hn * w1[j][e];
        }
    l0 = wave_sum(l0); l1 = wave_sum(l1);
    if (lane == 0) { red[1][w] = l0; red[2][w] = l1; }
    __syncthreads();
    if (tid == 0) {
        const float a = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]), b = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        logits[2 * row] = a; logits[2 * row + 1] = b;
        if (dec) dec[row] = b > a ? 1 : 0;                 // tie -> index 0, as gate_decide_kernel
    }
}
int sm_gate_tail(const float* x, int M, int D, int ldx, const float* gamma, float eps, const float* head_w_f32, float* logits, int32_t* decisions, void* stream) {
    SM_REQUIRE(x && gamma && head_w_f32 && logits && M > 0 && D > 0 && D <= 8192 && (D & 3) == 0 && (ldx & 3) == 0, "sm_gate_tail: bad args (D <= 8192, %% 4)");
    gate_tail_kernel<<<M, 256, 0, (hipStream_t)stream>>>(x, D, ldx, gamma, eps, head_w_f32, logits, decisions);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
// rows of a packed 16-bit weight image (sm_pack_weight layout) widened to fp32 row-major [N][K]
__global__ void unpack_rows_f32_kernel(const bf16_t* __restrict__ wp, int N, int K, int KS, int f16, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * K) return;
    const int n = t / K, k = t - n * K;
    const uint16_t raw = ((const uint16_t*)wp)[packed_index(n, k, KS)];
    out[t] = f16 ? (float)__builtin_bit_cast(_Float16, raw) : __uint_as_float((uint32_t)raw << 16);
}
int sm_unpack_rows_f32(const void* wp, int N, int K, int f16, float* out, void* stream) {
    SM_REQUIRE(wp && out && N > 0 && K > 0, "sm_unpack_rows_f32: bad args");
    unpack_rows_f32_kernel<<<cdiv(N * K, 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)wp, N, K, (K + 31) / 32, f16, out);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

__global__ void gate_decide_kernel(const float* lg, int M, int32_t* dec) {
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) dec[m] = lg[2 * m + 1] > lg[2 * m] ? 1 : 0;   // softmax is monotone; tie -> index 0
}
extern "C" int sm_gate_decide(const float* logits, int M, int32_t* decision, void* stream) {
    SM_REQUIRE(logits && decision && M > 0, "sm_gate_decide: bad args");
    gate_decide_kernel<<<cdiv(M, 64), 64, 0, (hipStream_t)stream>>>(logits, M, decision);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// ------------------------------------------------------------------------------------------------ LLM glue
// f16: the 16-bit tables / outputs of the LLM glue kernels below are IEEE fp16 instead of bf16 (llm_fp16 mode)
// ids >= vocab (tokens ADDED to the tokenizer after the checkpoint was written, e.g. <im_patch>; the reference grows the table by
// fresh rows there, builder.py:186-191) and frame indices >= n_tok read as ZERO rows instead of memory behind the buffers
// (vocab / n_tok <= 0: unchecked, the operator-level entry point)
__global__ void embed_splice_kernel(const int32_t* ids, int n, const bf16_t* table, const float* tokens, int D, float* out, int f16,
                                    int vocab, int n_tok) {
    int row = blockIdx.x;
    int id = ids[row];
    const bool text = id >= 0;
    const bool ok = text ? (vocab <= 0 || id < vocab) : (n_tok <= 0 || -id - 1 < n_tok);
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float v = 0.f;
        if (ok) v = text ? (f16 ? h2f(table[(size_t)id * D + c]) : bf2f(table[(size_t)id * D + c])) : tokens[(size_t)(-id - 1) * D + c];
        out[(size_t)row * D + c] = v;
    }
}
int sm_embed_splice_ex(const int32_t* ids, int n, const void* table, const float* tokens, int D, float* out, int f16, int vocab,
                       int n_tok, void* stream) {
    SM_REQUIRE(ids && table && out && n > 0, "sm_embed_splice: bad args");
    embed_splice_kernel<<<n, 256, 0, (hipStream_t)stream>>>(ids, n, (const bf16_t*)table, tokens, D, out, f16, vocab, n_tok);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
extern "C" int sm_embed_splice(const int32_t* ids, int n, const void* table, const float* tokens, int D, float* out, void* stream) {
    return sm_embed_splice_ex(ids, n, table, tokens, D, out, 0, 0, 0, stream);
}

// qkv fp32 [n][(H+2KV)*dh]; rotate_half: out[j] = x[j] cos - x[j+h] sin ; out[j+h] = x[j+h] cos + x[j] sin.
// grid (token, head slot): slots 0..H-1 = q heads, H..H+KV-1 = k heads (rotated, appended), H+KV.. = v heads (appended
// transposed); one 64-thread block each so a single decode token still spreads over H+2KV blocks.
__global__ __launch_bounds__(64) void rope_kv_kernel(const float* __restrict__ qkv, int pos0, int H, int KV, int dh,
                                                     const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                                     bf16_t* __restrict__ q, bf16_t* __restrict__ kc, bf16_t* __restrict__ vtc,
                                                     int S_max, int f16) {
    const int t = blockIdx.x, hd = blockIdx.y;
    const int pos = pos0 + t;
    const int half = dh >> 1;
    const float* x = qkv + (size_t)t * (H + 2 * KV) * dh + (size_t)hd * dh;
    if (hd < H + KV) {
        for (int j = threadIdx.x; j < half; j += 64) {
            const float c = cos_tab[(size_t)pos * half + j], s = sin_tab[(size_t)pos * half + j];
            const float a = x[j], b = x[j + half];
            const float o0 = a * c - b * s, o1 = b * c + a * s;
            bf16_t* dst = hd < H ? q + ((size_t)t * H + hd) * dh : kc + ((size_t)pos * KV + (hd - H)) * dh;
            dst[j] = (bf16_t)cvt16_rt(o0, f16);
            dst[j + half] = (bf16_t)cvt16_rt(o1, f16);
        }
    } else {
        const int kh = hd - H - KV;
        for (int e = threadIdx.x; e < dh; e += 64) vtc[((size_t)kh * dh + e) * S_max + pos] = (bf16_t)cvt16_rt(x[e], f16);
    }
}
// The same for a CHUNK of rows (prefill), head_dim 128: one 256-thread block per (64 tokens, head).  q / k heads: a thread rotates 16 (d, d + 64) pairs of one
// token with 16-byte loads (values, cos, sin) and 16-byte stores; v heads: the 64 x 128 tile is transposed through LDS so that the V^T cache is written as 64
// consecutive positions of one dim per wave-store (the row kernel above writes ONE 2-byte element per lane with a stride of S_max: 36 us per layer at 2048
// tokens, 2.1 TB/s).  Same arithmetic (a c - b s, b c + a s in fp32, one rounding), bit-identical outputs.
__global__ __launch_bounds__(256) void rope_kv_tile_kernel(const float* __restrict__ qkv, int n, int pos0, int H, int KV,
                                                           const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                                           bf16_t* __restrict__ q, bf16_t* __restrict__ kc, bf16_t* __restrict__ vtc, int S_max, int f16) {
    constexpr int DH = 128, HALF = 64;
    __shared__ bf16_t tile[DH][64 + 2];
    const int t0 = blockIdx.x * 64, hd = blockIdx.y, tid = threadIdx.x;
    const int tl = tid >> 2, part = tid & 3, t = t0 + tl;
    const int ld = (H + 2 * KV) * DH;
    if (hd < H + KV) {
        if (t >= n) return;
        const int pos = pos0 + t;
        const float* x = qkv + (size_t)t * ld + (size_t)hd * DH + part * 16;
        const float* cp = cos_tab + (size_t)pos * HALF + part * 16;
        const float* sp = sin_tab + (size_t)pos * HALF + part * 16;
        bf16_t* dst = (hd < H ? q + ((size_t)t * H + hd) * DH : kc + ((size_t)pos * KV + (hd - H)) * DH) + part * 16;
        uint32_t o0[8], o1[8];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const f32x4 a = *(const f32x4*)(x + v * 4), b = *(const f32x4*)(x + HALF + v * 4);
            const f32x4 c = *(const f32x4*)(cp + v * 4), s_ = *(const f32x4*)(sp + v * 4);
            float r0[4], r1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { r0[e] = a[e] * c[e] - b[e] * s_[e]; r1[e] = b[e] * c[e] + a[e] * s_[e]; }
            o0[v * 2] = pack16_rt(r0[0], r0[1], f16); o0[v * 2 + 1] = pack16_rt(r0[2], r0[3], f16);
            o1[v * 2] = pack16_rt(r1[0], r1[1], f16); o1[v * 2 + 1] = pack16_rt(r1[2], r1[3], f16);
        }
        *(u32x4*)(dst) = u32x4{o0[0], o0[1], o0[2], o0[3]};
        *(u32x4*)(dst + 8) = u32x4{o0[4], o0[5], o0[6], o0[7]};
        *(u32x4*)(dst + HALF) = u32x4{o1[0], o1[1], o1[2], o1[3]};
        *(u32x4*)(dst + HALF + 8) = u32x4{o1[4], o1[5], o1[6], o1[7]};
        return;
    }
    const int kh = hd - H - KV;
    if (t < n) {
        const float* x = qkv + (size_t)t * ld + (size_t)hd * DH + part * 32;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const f32x4 a = *(const f32x4*)(x + v * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[part * 32 + v * 4 + e][tl] = (bf16_t)cvt16_rt(a[e], f16);
        }
    }
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6;
    if (t0 + lane < n) {
        bf16_t* dstv = vtc + ((size_t)kh * DH + w * 32) * S_max + pos0 + t0 + lane;
#pragma unroll 8
        for (int e = 0; e < 32; ++e) dstv[(size_t)e * S_max] = tile[w * 32 + e][lane];
    }
}
int sm_rope_kv_append_ex(const float* qkv, int n, int pos0, int H, int KV, int dh, const float* cos_tab,
                         const float* sin_tab, void* q, void* kcache, void* vtcache, int S_max, int f16, void* stream) {
    SM_REQUIRE(qkv && q && cos_tab && sin_tab && kcache && vtcache && n > 0 && pos0 >= 0 && pos0 + n <= S_max, "sm_rope_kv_append: bad args (pos0=%d n=%d S_max=%d)", pos0, n, S_max);
    static int tile_on = -1;                      // SM_ROPE_TILE=0: the row kernel for every n (A/B)
    if (tile_on < 0) { const char* e = getenv("SM_ROPE_TILE"); tile_on = e ? atoi(e) : 1; }
    if (tile_on && dh == 128 && n >= 32 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)q & 15) == 0 && ((uintptr_t)kcache & 15) == 0 && ((uintptr_t)cos_tab & 15) == 0 &&
        ((uintptr_t)sin_tab & 15) == 0) {
        rope_kv_tile_kernel<<<dim3((n + 63) / 64, H + 2 * KV), 256, 0, (hipStream_t)stream>>>(qkv, n, pos0, H, KV, cos_tab, sin_tab, (bf16_t*)q, (bf16_t*)kcache,
                                                                                            (bf16_t*)vtcache, S_max, f16);
        SM_LAUNCH_CHECK();
        return SM_OK;
    }
    rope_kv_kernel<<<dim3(n, H + 2 * KV), 64, 0, (hipStream_t)stream>>>(qkv, pos0, H, KV, dh, cos_tab, sin_tab, (bf16_t*)q,
                                                                        (bf16_t*)kcache, (bf16_t*)vtcache, S_max, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
extern "C" int sm_rope_kv_append(const float* qkv, int n, int pos0, int H, int KV, int dh, const float* cos_tab,
                                 const float* sin_tab, void* q, void* kcache, void* vtcache, int S_max, void* stream) {
    return sm_rope_kv_append_ex(qkv, n, pos0, H, KV, dh, cos_tab, sin_tab, q, kcache, vtcache, S_max, 0, stream);
}

// the same for S streams x one token each: row t = stream t at ITS position, appended to ITS caches
template <class SEG>
__global__ __launch_bounds__(64) void rope_kv_seg_kernel(const float* __restrict__ qkv, int H, int KV, int dh,
                                                         const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                                         bf16_t* __restrict__ q, SEG seg, int S_max, int f16, int nslab = 0, size_t sstride = 0) {
    const int t = blockIdx.x, hd = blockIdx.y;
    const int pos = seg_pos(seg, t);
    bf16_t* __restrict__ kc = (bf16_t*)seg.kc[t];
    bf16_t* __restrict__ vtc = (bf16_t*)seg.vtc[t];
    const int half = dh >> 1;
    const float* x0 = qkv + (size_t)t * (H + 2 * KV) * dh + (size_t)hd * dh;
    // nslab > 0: qkv holds the q|k|v product's UNSUMMED split-K slabs ([nslab][rows][(H + 2 KV) dh], sm_linear_leave_slabs): summed here in slab order, the
    // arithmetic of the reduce launch this replaces
    struct Row { const float* p; int n; size_t st;
                 __device__ float operator[](int i) const { if (n <= 0) return p[i]; float v = 0.f; for (int s_ = 0; s_ < n; ++s_) v += p[(size_t)s_ * st + i]; return v; } };
    const Row x = {x0, nslab, sstride};
    if (hd < H + KV) {
        for (int j = threadIdx.x; j < half; j += 64) {
            const float c = cos_tab[(size_t)pos * half + j], s = sin_tab[(size_t)pos * half + j];
            const float a = x[j], b = x[j + half];
            const float o0 = a * c - b * s, o1 = b * c + a * s;
            bf16_t* dst = hd < H ? q + ((size_t)t * H + hd) * dh : kc + ((size_t)pos * KV + (hd - H)) * dh;
            dst[j] = (bf16_t)cvt16_rt(o0, f16);
            dst[j + half] = (bf16_t)cvt16_rt(o1, f16);
        }
    } else {
        const int kh = hd - H - KV;
        for (int e = threadIdx.x; e < dh; e += 64) vtc[((size_t)kh * dh + e) * S_max + pos] = (bf16_t)cvt16_rt(x[e], f16);
    }
}
int sm_rope_kv_append_seg(const float* qkv, int S, int H, int KV, int dh, const float* cos_tab, const float* sin_tab, void* q,
                          const SmDecodeSeg& seg, int S_max, int f16, void* stream) {
    SM_REQUIRE(qkv && q && cos_tab && sin_tab && S > 0 && S <= SM_MAX_SEG, "sm_rope_kv_append_seg: bad args");
    rope_kv_seg_kernel<SmDecodeSeg><<<dim3(S, H + 2 * KV), 64, 0, (hipStream_t)stream>>>(qkv, H, KV, dh, cos_tab, sin_tab, (bf16_t*)q, seg, S_max, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
int sm_rope_kv_append_seg_big(const float* qkv, int S, int H, int KV, int dh, const float* cos_tab, const float* sin_tab, void* q,
                              const SmDecodeSegBig& seg, int S_max, int f16, void* stream) {
    SM_REQUIRE(qkv && q && cos_tab && sin_tab && S > 0 && S <= SM_BIG_SEG, "sm_rope_kv_append_seg_big: bad args");
    rope_kv_seg_kernel<SmDecodeSegBig><<<dim3(S, H + 2 * KV), 64, 0, (hipStream_t)stream>>>(qkv, H, KV, dh, cos_tab, sin_tab, (bf16_t*)q, seg, S_max, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

int sm_rope_kv_append_seg_tab(const float* qkv, int S, int H, int KV, int dh, const float* cos_tab, const float* sin_tab, void* q,
                              const SmDecodeSegTab& tab, int S_max, int f16, void* stream, int nslab, size_t slab_stride) {
    SM_REQUIRE(qkv && q && cos_tab && sin_tab && S > 0 && tab.kc && tab.vtc && tab.pos0 && nslab >= 0 && nslab <= 64, "sm_rope_kv_append_seg_tab: bad args");
    rope_kv_seg_kernel<SmDecodeSegTab><<<dim3(S, H + 2 * KV), 64, 0, (hipStream_t)stream>>>(qkv, H, KV, dh, cos_tab, sin_tab, (bf16_t*)q, tab, S_max, f16, nslab, slab_stride);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// row t: emit stream t's pending token (out_rows.p[t][col]) and gather its embedding
template <typename PACK>
__global__ void embed_tokens_seg_kernel(PACK tok, const bf16_t* __restrict__ table, int D, float* __restrict__ out,
                                        PACK out_rows, int col, int f16) {
    const int t = blockIdx.x;
    const int id = *tok.p[t];
    if (threadIdx.x == 0) out_rows.p[t][col] = id;
    for (int c = threadIdx.x; c < D; c += blockDim.x) out[(size_t)t * D + c] = f16 ? h2f(table[(size_t)id * D + c]) : bf2f(table[(size_t)id * D + c]);
}
int sm_embed_tokens_seg(const SmTokPtrs& tok, int S, const void* table, int D, float* out, const SmTokPtrs& out_rows, int col,
                        int f16, void* stream) {
    SM_REQUIRE(table && out && S > 0 && S <= SM_MAX_SEG, "sm_embed_tokens_seg: bad args");
    embed_tokens_seg_kernel<SmTokPtrs><<<S, 256, 0, (hipStream_t)stream>>>(tok, (const bf16_t*)table, D, out, out_rows, col, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
int sm_embed_tokens_seg_big(const SmTokPtrsBig& tok, int S, const void* table, int D, float* out, const SmTokPtrsBig& out_rows, int col,
                            int f16, void* stream) {
    SM_REQUIRE(table && out && S > 0 && S <= SM_BIG_SEG, "sm_embed_tokens_seg_big: bad args");
    embed_tokens_seg_kernel<SmTokPtrsBig><<<S, 256, 0, (hipStream_t)stream>>>(tok, (const bf16_t*)table, D, out, out_rows, col, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

__global__ void swiglu_kernel(const float* __restrict__ gu, int M, int F, bf16_t* __restrict__ out, int f16) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t tot = (size_t)M * (F >> 2);
    if (t >= tot) return;
    int m = (int)(t / (F >> 2)), c = (int)(t % (F >> 2)) * 4;
    f32x4 g = *(const f32x4*)(gu + (size_t)m * 2 * F + c);
    f32x4 u = *(const f32x4*)(gu + (size_t)m * 2 * F + F + c);
    *(u32x2*)(out + (size_t)m * F + c) = u32x2{pack16_rt(siluf_(g[0]) * u[0], siluf_(g[1]) * u[1], f16),
                                              pack16_rt(siluf_(g[2]) * u[2], siluf_(g[3]) * u[3], f16)};
}
int sm_swiglu_ex(const float* gu, int M, int F, void* out, int f16, void* stream) {
    SM_REQUIRE(gu && out && M > 0 && F % 4 == 0, "sm_swiglu: bad args");
    size_t tot = (size_t)M * (F / 4);
    swiglu_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(gu, M, F, (bf16_t*)out, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
extern "C" int sm_swiglu(const float* gu, int M, int F, void* out, void* stream) { return sm_swiglu_ex(gu, M, F, out, 0, stream); }

__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ lg, int V, int32_t* out) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        float t = lg[v];
        if (t > best) { best = t; idx = v; }     // strictly greater: lowest index wins within a thread
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(idx, o, 64);
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { bv[w] = best; bi[w] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
            if (bv[k] > best || (bv[k] == best && bi[k] < idx)) { best = bv[k]; idx = bi[k]; }
        *out = idx;
    }
}
// f1: per-row softmax cross-entropy + argmax; one 256-thread block per row, two passes over the (L2-resident) row
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ lg, int V, int ld, const int32_t* __restrict__ labels,
                                                            int ignore, float* __restrict__ nll, int32_t* __restrict__ amax) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const float* row = lg + (size_t)blockIdx.x * ld;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        const float t = row[v];
        if (t > best) { best = t; idx = v; }          // strictly greater: the lowest index wins inside a thread
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    if (lane == 0) { sv[w] = best; si[w] = idx; }
    __syncthreads();
    best = sv[0]; idx = si[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (sv[k] > best || (sv[k] == best && si[k] < idx)) { best = sv[k]; idx = si[k]; }
    __syncthreads();
    float sum = 0.f;
    for (int v = tid; v < V; v += 256) sum += __expf(row[v] - best);
    sum = wave_sum(sum);
    if (lane == 0) sv[w] = sum;
    __syncthreads();
    if (tid == 0) {
        if (amax) amax[blockIdx.x] = idx;
        if (nll) {
            const int lab = labels ? labels[blockIdx.x] : ignore;
            float out = 0.f;
            if (lab != ignore && lab >= 0 && lab < V) out = __logf((sv[0] + sv[1]) + (sv[2] + sv[3])) + best - row[lab];
            nll[blockIdx.x] = out;
        }
    }
}
extern "C" int sm_cross_entropy(const float* logits, int n, int V, int ld, const int32_t* labels, int ignore_index, float* nll,
                                int32_t* argmax, void* stream) {
    SM_REQUIRE(logits && n > 0 && V > 0 && ld >= V && (nll || argmax) && (!nll || labels), "sm_cross_entropy: bad args");
    cross_entropy_kernel<<<n, 256, 0, (hipStream_t)stream>>>(logits, V, ld, labels, ignore_index, nll, argmax);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// cosine similarity of every row of x [T][D] (fp32, row stride ld) with one reference row: one wave per row, 16-byte loads, fp32 sums.
// torch.nn.functional.cosine_similarity's arithmetic: x . r / (max(|x|, eps) * max(|r|, eps)), eps = 1e-8
// (videollama2_arch.py:603-611 ranks the frame tokens of a clip by it: "similarity" sampling)
__global__ __launch_bounds__(256) void cosine_rows_kernel(const float* __restrict__ x, int T, int D, int ld, const float* __restrict__ ref,
                                                          float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= T) return;
    const float* xr = x + (size_t)row * ld;
    float dot = 0.f, xx = 0.f, rr = 0.f;
    const bool vec = ((ld | D) & 3) == 0 && (((uintptr_t)x | (uintptr_t)ref) & 15) == 0;
    if (vec) {
        for (int k = lane * 4; k < D; k += 256) {
            const f32x4 a = *(const f32x4*)(xr + k), b = *(const f32x4*)(ref + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) { dot = __builtin_fmaf(a[j], b[j], dot); xx = __builtin_fmaf(a[j], a[j], xx); rr = __builtin_fmaf(b[j], b[j], rr); }
        }
    } else {
        for (int k = lane; k < D; k += 64) {
            const float a = xr[k], b = ref[k];
            dot = __builtin_fmaf(a, b, dot); xx = __builtin_fmaf(a, a, xx); rr = __builtin_fmaf(b, b, rr);
        }
    }
    dot = wave_sum(dot); xx = wave_sum(xx); rr = wave_sum(rr);
    if (lane == 0) out[row] = dot / (fmaxf(sqrtf(xx), 1e-8f) * fmaxf(sqrtf(rr), 1e-8f));
}
extern "C" int sm_cosine_rows(const float* x, int T, int D, int ld, const float* ref, float* out, void* stream) {
    SM_REQUIRE(x && ref && out && T > 0 && D > 0 && ld >= D, "sm_cosine_rows: bad args");
    cosine_rows_kernel<<<cdiv(T, 4), 256, 0, (hipStream_t)stream>>>(x, T, D, ld, ref, out);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// one block per row: greedy token of row t into stream t's own pending-token word
template <typename PACK>
__global__ __launch_bounds__(1024) void argmax_rows_seg_kernel(const float* __restrict__ lg, int V, int ld, PACK out) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const float* row = lg + (size_t)blockIdx.x * ld;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        float t = row[v];
        if (t > best) { best = t; idx = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(idx, o, 64);
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { bv[w] = best; bi[w] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
            if (bv[k] > best || (bv[k] == best && bi[k] < idx)) { best = bv[k]; idx = bi[k]; }
        *out.p[blockIdx.x] = idx;
    }
}
int sm_argmax_rows_seg(const float* logits, int S, int V, int ld, const SmTokPtrs& out, void* stream) {
    SM_REQUIRE(logits && S > 0 && S <= SM_MAX_SEG && V > 0, "sm_argmax_rows_seg: bad args");
    argmax_rows_seg_kernel<SmTokPtrs><<<S, 1024, 0, (hipStream_t)stream>>>(logits, V, ld, out);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
int sm_argmax_rows_seg_big(const float* logits, int S, int V, int ld, const SmTokPtrsBig& out, void* stream) {
    SM_REQUIRE(logits && S > 0 && S <= SM_BIG_SEG && V > 0, "sm_argmax_rows_seg_big: bad args");
    argmax_rows_seg_kernel<SmTokPtrsBig><<<S, 1024, 0, (hipStream_t)stream>>>(logits, V, ld, out);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

extern "C" int sm_argmax(const float* logits, int V, int32_t* out, void* stream) {
    SM_REQUIRE(logits && out && V > 0, "sm_argmax: bad args");
    argmax_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(logits, V, out);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
