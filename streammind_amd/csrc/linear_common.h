// Shared device code of the linear kernels: argument block, generic 4-wide epilogue, LDS-DMA helper.
#pragma once
#include "common.h"
#include "host.h"

// ------------------------------------------------------------------------------------------------ epilogue
struct LinArgs {
    const bf16x8* w;
    const bf16x8* w2;
    int N, K, KS, NRG;
    const void* x;
    int M, ldx;
    const float* bias;
    int act;
    const float* residual;
    int ldr;
    float* out_f32;
    bf16_t* out_bf16;
    int ldo, ldo_bf16;
    int remap_in, remap_out, remap_off;
    bf16_t* vt;
    int vt_n0, vt_S, vt_dh, vt_ld;
    const float* wscale;        // fp8 weights (skinny path): per-output-row dequantisation scale, applied to the accumulator
    const float* wscale2;       //   ... of the second (dual) weight
    const float* ngamma;        // fused RMSNorm of the activations (weight-streaming path): gamma [K], or nullptr
    float neps;
    int f16;                    // 16-bit operands AND 16-bit outputs are IEEE fp16 instead of bf16 (tiled GEMM path only)
    int xr_sh, xr_dh_sh;        // repeated column groups of x (sm_linear_t.x_rep): log2(x_rep) (0 = off), log2(x_rep_dh)
    // LayerNorm folding (sm_linear_t.fold_*; 256 x 256 tile kernels only).  Producer (fp32 + residual outputs): out_bf16 holds 16-bit(o * fold_og[n])
    // and fold_ostats[(m * (N / 256) + tile_n) * 2 + {0, 1}] the row's (sum, sum of squares) over the tile's 256 columns.  Consumer (16-bit outputs):
    // the accumulators S of the raw scaled rows become rstd[m] * (S - mu[m] * fold_ig[n]) + fold_ic[n], mu / rstd from row m's fold_itiles partial sums.
    const float* fold_og;
    float* fold_ostats;
    const float* fold_istats;
    const float* fold_ig;
    const float* fold_ic;
    int fold_itiles;
    float fold_invd, fold_eps;
    int n_band;           // 128 x 128 kernel's tile order: 0 = an XCD's band of tile ids walks n fastest (whole ROW tiles per XCD: X stays in its L2), 1 = m fastest (whole COLUMNS of
                          // tiles per XCD: the row tiles that multiply the same W tile share an L2 -- the weight-heavy shapes, fewer row tiles than column tiles)
};

// (-mu * rstd, rstd) of row m from the partial sums the producing GEMM left: tiles in order (deterministic), E[x^2] - mu^2
static __device__ __forceinline__ f32x2 fold_row_stats(const LinArgs& a, int m) {
    if (m >= a.M) m = a.M - 1;
    float s1, s2;
    if (a.fold_itiles == 4) {                    // the tower's width (1024): one row's partials are 32 contiguous bytes
        const f32x4* p = (const f32x4*)(a.fold_istats + (size_t)m * 8);
        const f32x4 u = p[0], v = p[1];
        s1 = ((u[0] + u[2]) + v[0]) + v[2];
        s2 = ((u[1] + u[3]) + v[1]) + v[3];
    } else {
        s1 = s2 = 0.f;
        for (int t = 0; t < a.fold_itiles; ++t) {
            const f32x2 v = *(const f32x2*)(a.fold_istats + ((size_t)m * a.fold_itiles + t) * 2);
            s1 += v[0]; s2 += v[1];
        }
    }
    const float mu = s1 * a.fold_invd;
    const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(s2, a.fold_invd, a.fold_eps) - mu * mu);
    return f32x2{-mu * rstd, rstd};
}

// column of the stored x row that operand column k reads (k a multiple of 8: a fragment's 8 columns stay inside one group)
static __device__ __forceinline__ int xcol(const LinArgs& a, int k) {
    return a.xr_sh ? (((k >> (a.xr_sh + a.xr_dh_sh)) << a.xr_dh_sh) | (k & ((1 << a.xr_dh_sh) - 1))) : k;
}

static __device__ __noinline__ float apply_act_rt(float v, int act) { return apply_act(v, act); }

// one lane's 4 consecutive outputs (n0..n0+3) of row m
static __device__ __forceinline__ void store4(const LinArgs& a, int m, int n0, f32x4 v, const f32x4* v2) {
    if (m >= a.M || n0 >= a.N) return;
    int orow = m, rrow = m;
    if (a.remap_in > 0) {
        int q = m / a.remap_in, r = m - q * a.remap_in;
        orow = q * a.remap_out + a.remap_off + r;
        rrow = a.remap_off + r;
    }
    const bool full = (n0 + 3 < a.N);
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = v[r];
        const bool inb = full || n0 + r < a.N;
        if (a.wscale && inb) t *= a.wscale[n0 + r];
        if (a.bias && inb) t += a.bias[n0 + r];
        if (v2) t = siluf_(t) * ((*v2)[r] * ((a.wscale2 && inb) ? a.wscale2[n0 + r] : 1.0f));
        else t = apply_act_rt(t, a.act);
        if (a.residual && (full || n0 + r < a.N)) t += a.residual[(size_t)rrow * a.ldr + n0 + r];
        o[r] = t;
    }
    if (a.out_f32) {
        float* p = a.out_f32 + (size_t)orow * a.ldo + n0;
        if (full && ((a.ldo & 3) == 0)) {
            *(f32x4*)p = f32x4{o[0], o[1], o[2], o[3]};
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n0 + r < a.N) p[r] = o[r];
        }
    }
    if (a.vt && n0 >= a.vt_n0) {
        int b = m / a.vt_S, s = m - b * a.vt_S;
        int nh = (a.N - a.vt_n0) / a.vt_dh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int c = n0 + r - a.vt_n0;
            if (n0 + r < a.N) {
                int h = c / a.vt_dh, d = c - h * a.vt_dh;
                a.vt[((size_t)(b * nh + h) * a.vt_dh + d) * a.vt_ld + s] = (bf16_t)cvt16_rt(o[r], a.f16);
            }
        }
    } else if (a.out_bf16) {
        bf16_t* p = a.out_bf16 + (size_t)orow * a.ldo_bf16 + n0;
        if (full && ((a.ldo_bf16 & 3) == 0)) {
            *(u32x2*)p = u32x2{pack16_rt(o[0], o[1], a.f16), pack16_rt(o[2], o[3], a.f16)};
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n0 + r < a.N) p[r] = (bf16_t)cvt16_rt(o[r], a.f16);
        }
    }
}


typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// 16 B per lane global -> LDS DMA; the LDS destination is wave-uniform base + lane*16 (1 KiB per wave-instruction)
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

// LayerNorm / RMSNorm of the finished output row riding behind a product (sm_linear_t.post_ln_*)
struct PostLn { const float* gamma; const float* beta; float eps; bf16_t* out; int ldo; float* out_f32; int act; };
int launch_gemm256(const LinArgs& a, int act, int bn, hipStream_t st, int S = 1);      // gemm256.hip (256 x bn tile; a.f16 selects the fp16 build; S > 1: split-K slabs [S][M][ldo] of raw accumulators in out_f32)
int launch_wstream(const LinArgs& a, hipStream_t st, float* ws, int S, int ksl);      // wstream.hip (33..128 rows: weights straight into register rings, X through LDS by a loader wave)
int launch_gemm_fp8(LinArgs& a, hipStream_t st);                            // gemm_fp8.hip (fp8 x fp8 MFMA, activations quantised per row)
int splitk_workspace(hipStream_t st, size_t bytes, float** out);            // linear.hip: per-HIP-stream fp32 slabs of the split-K kernels
int launch_splitk_reduce(const LinArgs& a, const float* ws, int S, int ldw, hipStream_t st);     // ... their fixed-order sum + the real epilogue
void release_stream_workspaces(hipStream_t st);                             // linear.hip: free the per-stream slabs of a stream about to be destroyed
