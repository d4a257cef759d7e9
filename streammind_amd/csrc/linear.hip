// Linear ops for gfx950: weight packer, weight-streaming "skinny" kernel (M <= 16) and the LDS-tiled MFMA GEMM.
//
// Both compute kernels use v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand and the activations as
// the B operand, i.e. they produce D[i = n][j = m] (the transposed output tile).  In that orientation a lane
// holds 4 consecutive n for one m (C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg), so bias loads and
// output stores are 8/16-byte vectors along the contiguous dimension of the row-major output.
#include <stdlib.h>

#include "linear_common.h"

// ------------------------------------------------------------------------------------------------ pack
__global__ void pack_weight_kernel(const bf16_t* __restrict__ w, int N, int K, int ldw, u32x4* __restrict__ out,
                                   int KS, size_t total_chunks) {
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total_chunks) return;
    int lane = (int)(c & 63);
    size_t rest = c >> 6;
    int ks = (int)(rest % KS);
    int rg = (int)(rest / KS);
    int n = rg * 16 + (lane & 15);
    int k0 = ks * 32 + (lane >> 4) * 8;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t lo = 0, hi = 0;
        if (n < N) {
            if (k0 + 2 * j < K) lo = w[(size_t)n * ldw + k0 + 2 * j];
            if (k0 + 2 * j + 1 < K) hi = w[(size_t)n * ldw + k0 + 2 * j + 1];
        }
        v[j] = lo | (hi << 16);
    }
    u32x4 o = {v[0], v[1], v[2], v[3]};
    out[c] = o;
}

extern "C" size_t sm_packed_elems(int N, int K) {
    return (size_t)((N + 15) / 16) * ((K + 31) / 32) * 512;
}

// KS_out >= ceil(K/32): number of k-steps of the packed image (extra steps are zero-filled)
int sm_pack_weight_ks(const void* w, int N, int K, int ldw, int KS, void* out, void* stream) {
    SM_REQUIRE(w && out && N > 0 && K > 0 && ldw >= K && KS * 32 >= K, "sm_pack_weight: bad args N=%d K=%d ldw=%d KS=%d", N, K, ldw, KS);
    size_t chunks = (size_t)((N + 15) / 16) * KS * 64;
    int blocks = (int)((chunks + 255) / 256);
    pack_weight_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)w, N, K, ldw, (u32x4*)out, KS, chunks);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

extern "C" int sm_pack_weight(const void* w, int N, int K, int ldw, void* out, void* stream) {
    return sm_pack_weight_ks(w, N, K, ldw, (K + 31) / 32, out, stream);
}

// fp32 -> bf16 hi (+ lo = bf16(x - hi), the next 8 mantissa bits) with the hardware RNE conversion (v_cvt_pk_bf16_f32)
template <bool SPLIT, bool F16 = false>
__device__ __forceinline__ void split_x(f32x4 a, f32x4 b, bf16x8& hi, bf16x8& lo) {
    const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if constexpr (F16) {                                   // IEEE fp16 operands (llm_fp16 / vit_fp16): same storage, other rounding
        f16x8 Hh, Lh;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const _Float16 h = (_Float16)f[j];
            Hh[j] = h;
            if (SPLIT) Lh[j] = (_Float16)(f[j] - (float)h);
        }
        hi = __builtin_bit_cast(bf16x8, Hh);
        if (SPLIT) lo = __builtin_bit_cast(bf16x8, Lh);
        return;
    }
    bf16x8 H, L;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)f[j];
        H[j] = h;
        if (SPLIT) L[j] = (__bf16)(f[j] - (float)h);
    }
    hi = H;
    if (SPLIT) lo = L;
}

// x fragment of the skinny kernels: 8 consecutive k of row m, as bf16 (hi) and optionally the bf16 residual (lo)
template <bool XF32, bool SPLIT, bool F16 = false>
__device__ __forceinline__ void load_x(const char* xrow, int k, bool valid, bf16x8& hi, bf16x8& lo) {
    if (XF32) {
        f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        if (valid) {
            a = *(const f32x4*)(xrow + (size_t)k * 4);
            b = *(const f32x4*)(xrow + (size_t)k * 4 + 16);
        }
        split_x<SPLIT, F16>(a, b, hi, lo);
    } else {
        union { bf16x8 v; u32x4 u; } H;
        H.u = u32x4{0, 0, 0, 0};
        if (valid) H.u = *(const u32x4*)(xrow + (size_t)k * 2);
        hi = H.v;
    }
}

// ------------------------------------------------------------------------------------------------ fp8 weights
// Weight-only fp8 (OCP e4m3, gfx950 v_cvt_pk_fp8_f32) for the HBM-bound weight-streaming path: per-output-row scale
// s[n] = max|W[n,:]| / 448, q = fp8(W / s).  Packed image: [N/16][ceil(KS/2)][lane][16 B], a lane's 16 bytes holding
// its 8 elements of k-step 2p followed by those of k-step 2p+1, so one 16-byte load feeds two MFMAs after an exact
// fp8 -> bf16 expansion in registers.  BASELINE config 5; opt-in (numerics differ from the bf16 checkpoint).
__global__ void rowscale_kernel(const bf16_t* __restrict__ w, int N, int K, int ldw, float* __restrict__ scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    float m = 0.f;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(bf2f(w[(size_t)row * ldw + k])));
    m = wave_max(m);
    if (lane == 0) scale[row] = m > 0.f ? m * (1.0f / 448.0f) : 1.0f;
}
__global__ void pack_fp8_kernel(const bf16_t* __restrict__ w, int N, int K, int ldw, const float* __restrict__ scale,
                                u32x4* __restrict__ out, int KSP, size_t total_chunks) {
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total_chunks) return;
    const int lane = (int)(c & 63);
    const size_t rest = c >> 6;
    const int kp = (int)(rest % KSP), rg = (int)(rest / KSP);
    const int n = rg * 16 + (lane & 15);
    const float inv = n < N ? 1.0f / scale[n] : 0.f;
    uint32_t o[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k0 = (2 * kp + h) * 32 + (lane >> 4) * 8;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (n < N && k0 + j < K) ? bf2f(w[(size_t)n * ldw + k0 + j]) * inv : 0.f;
        int r0 = 0, r1 = 0;
        r0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], r0, false);
        r0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], r0, true);
        r1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], r1, false);
        r1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], r1, true);
        o[2 * h] = (uint32_t)r0; o[2 * h + 1] = (uint32_t)r1;
    }
    out[c] = u32x4{o[0], o[1], o[2], o[3]};
}
extern "C" size_t sm_packed_fp8_bytes(int N, int K) {
    const int KS = (K + 31) / 32;
    return (size_t)((N + 15) / 16) * ((KS + 1) / 2) * 1024;
}
extern "C" int sm_quant_pack_weight_fp8(const void* w, int N, int K, int ldw, void* out, float* scale_out, void* stream) {
    SM_REQUIRE(w && out && scale_out && N > 0 && K > 0 && ldw >= K, "sm_quant_pack_weight_fp8: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int KSP = ((K + 31) / 32 + 1) / 2;
    rowscale_kernel<<<cdiv(N, 4), 256, 0, st>>>((const bf16_t*)w, N, K, ldw, scale_out);
    const size_t chunks = (size_t)((N + 15) / 16) * KSP * 64;
    pack_fp8_kernel<<<(unsigned)((chunks + 255) / 256), 256, 0, st>>>((const bf16_t*)w, N, K, ldw, scale_out, (u32x4*)out, KSP, chunks);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// 8 fp8 (two dwords) -> 8 bf16 with v_cvt_scalef32_pk_bf16_fp8 (gfx950): ONE instruction per pair (scale 1: every e4m3 value is
// exact in bf16) instead of v_cvt_pk_f32_fp8 + v_perm_b32 -- the conversion is what bounds the fp8 weight-streaming kernels
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(uint32_t lo, uint32_t hi) {
    union { bf16x8 v; bf16x2 p[4]; } r;
    r.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false);
    r.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    r.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false);
    r.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    return r.v;
}

// fp8 packed image -> the bf16 packed image the tiled kernels read, row scale folded in (bf16(q * s[n]), RNE): what M > 16
// rows do with fp8 weights (prefill chunks, teacher-forced evaluation).  One lane = the 16 bytes of two k-steps.
__global__ void dequant_fp8_kernel(const u32x4* __restrict__ in, const float* __restrict__ scale, int N, int KS, int KSP,
                                   u32x4* __restrict__ out, size_t total_chunks) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total_chunks) return;
    const int lane = (int)(c & 63);
    const size_t rest = c >> 6;
    const int kp = (int)(rest % KSP), rg = (int)(rest / KSP);
    const int n = rg * 16 + (lane & 15);
    const float sc = n < N ? scale[n] : 0.f;
    const u32x4 q = in[c];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ks = 2 * kp + h;
        if (ks >= KS) break;
        const uint32_t lo = h ? q[2] : q[0], hi = h ? q[3] : q[1];
        const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
        const f32x2 d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), e = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
        out[((size_t)rg * KS + ks) * 64 + lane] = u32x4{pack2bf(a[0] * sc, a[1] * sc), pack2bf(b[0] * sc, b[1] * sc),
                                                        pack2bf(d[0] * sc, d[1] * sc), pack2bf(e[0] * sc, e[1] * sc)};
    }
}

// ---- fused RMSNorm of the weight-streaming kernels (NORM): the activations are the raw fp32 residual stream.  Every block
// recomputes mean(x^2) of its rows from L2 while its first batch of weight loads is already in flight, writes the normalised
// rows bf16(gamma * (x * rstd)) -- sm_norm's own formula -- to LDS ONCE and reads its x fragments from there (re-reading x and
// gamma per k-step made the kernels texture-address bound).  norm_issue goes BEFORE the caller's first weight loads (vmcnt
// retires in order: waiting for the row must not wait for HBM); the partial sums cross the waves behind raw s_barriers
// (__syncthreads would drain vmcnt).  Wave w owns the f32x4 chunks {(u * WAVES + w) * 64 + lane}; the first four per lane
// (K <= 1024 * WAVES) stay in registers between the sum and the conversion.
template <int WAVES>
__device__ __forceinline__ void norm_issue(const LinArgs& a, int wave, int lane, f32x4 (&t)[4], f32x4 (&gm)[4]) {
    const int nv = a.K >> 2;
    const float* xr = (const float*)a.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = (u * WAVES + wave) * 64 + lane;
        t[u] = c < nv ? *(const f32x4*)(xr + (size_t)c * 4) : f32x4{0, 0, 0, 0};
        gm[u] = c < nv ? *(const f32x4*)(a.ngamma + (size_t)c * 4) : f32x4{0, 0, 0, 0};
    }
}
template <int WAVES, bool F16 = false>
__device__ __forceinline__ void norm_finish(const LinArgs& a, float* nsum, bf16_t* xs, int wave, int lane, f32x4 (&t)[4], const f32x4 (&gm)[4]) {
    const int nv = a.K >> 2;
    const float* xr = (const float*)a.x;
    for (int m = 0; m < a.M; ++m) {
        const float* xm = xr + (size_t)m * a.ldx;
        if (m) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = (u * WAVES + wave) * 64 + lane;
                t[u] = c < nv ? *(const f32x4*)(xm + (size_t)c * 4) : f32x4{0, 0, 0, 0};
            }
        }
        float sq = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) sq += t[u][0] * t[u][0] + t[u][1] * t[u][1] + t[u][2] * t[u][2] + t[u][3] * t[u][3];
        for (int c = (4 * WAVES + wave) * 64 + lane; c < nv; c += WAVES * 64) {      // K > 1024 * WAVES: the rest, chunk by chunk
            const f32x4 q = *(const f32x4*)(xm + (size_t)c * 4);
            sq += q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        }
        sq = wave_sum(sq);
        if (lane == 0) nsum[(m & 1) * WAVES + wave] = sq;            // two alternating slots of WAVES partials
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) tot += nsum[(m & 1) * WAVES + w];
        const float rs = rsqrtf(tot / (float)a.K + a.neps);
        bf16_t* xo = xs + (size_t)m * a.K;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = (u * WAVES + wave) * 64 + lane;
            if (c < nv) *(u32x2*)(xo + (size_t)c * 4) = u32x2{pack16<F16>(gm[u][0] * (t[u][0] * rs), gm[u][1] * (t[u][1] * rs)),
                                                               pack16<F16>(gm[u][2] * (t[u][2] * rs), gm[u][3] * (t[u][3] * rs))};
        }
        for (int c = (4 * WAVES + wave) * 64 + lane; c < nv; c += WAVES * 64) {
            const f32x4 q = *(const f32x4*)(xm + (size_t)c * 4), gq = *(const f32x4*)(a.ngamma + (size_t)c * 4);
            *(u32x2*)(xo + (size_t)c * 4) = u32x2{pack16<F16>(gq[0] * (q[0] * rs), gq[1] * (q[1] * rs)), pack16<F16>(gq[2] * (q[2] * rs), gq[3] * (q[3] * rs))};
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// one lane's share of the fused RoPE / KV-append epilogue: row `row` (= stream), head slot hs (q heads, then k heads, then v
// heads), dims d0..d0+3 in `lo` and d0+64..d0+67 in `hi` of a 128-wide head.  Same arithmetic as rope_kv_kernel (vecops.hip).
template <bool F16 = false>
static __device__ __forceinline__ void rope_store(const SmRopeEpi& re, int row, int hs, int d0, f32x4 lo, f32x4 hi) {
    const int pos = re.seg.pos[row];
    if (hs < re.H + re.KV) {
        const f32x4 c = *(const f32x4*)(re.cos_tab + (size_t)pos * 64 + d0), s = *(const f32x4*)(re.sin_tab + (size_t)pos * 64 + d0);
        float o0[4], o1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0[r] = lo[r] * c[r] - hi[r] * s[r];
            o1[r] = hi[r] * c[r] + lo[r] * s[r];
        }
        bf16_t* dst = hs < re.H ? (bf16_t*)re.q + ((size_t)row * re.H + hs) * 128 + d0
                                : (bf16_t*)re.seg.kc[row] + ((size_t)pos * re.KV + (hs - re.H)) * 128 + d0;
        *(u32x2*)dst = u32x2{pack16<F16>(o0[0], o0[1]), pack16<F16>(o0[2], o0[3])};
        *(u32x2*)(dst + 64) = u32x2{pack16<F16>(o1[0], o1[1]), pack16<F16>(o1[2], o1[3])};
    } else {
        bf16_t* vt = (bf16_t*)re.seg.vtc[row] + ((size_t)(hs - re.H - re.KV) * 128 + d0) * re.S_max + pos;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            vt[(size_t)r * re.S_max] = (bf16_t)cvt16<F16>(lo[r]);
            vt[(size_t)(64 + r) * re.S_max] = (bf16_t)cvt16<F16>(hi[r]);
        }
    }
}

// ---- fp8 weight streaming (M <= 16): same block / wave decomposition as skinny_kernel, the weights as fp8 pairs of k-steps.
// Wave w owns the 1 KiB chunks kp = w + c * WAVES of its row group and keeps D of them IN FLIGHT at all times (register ring:
// consume the oldest, issue the next into its registers -- the compiler's in-order vmcnt accounting gives the counted waits);
// the ragged rest is issued in one go behind the last full round.  Round 2's loop issued a batch, waited for all of it and
// fetched each x fragment under an exec-masked branch between the MFMAs (eight serial L2 round trips per batch: the fp8 down
// projection of a decode step took as long as the bf16 one).  The x fragments now ride the ring (unmasked loads, row clamped,
// zeroed by a select), and everything the epilogue needs -- the row scales, the residual rows -- is fetched FIRST, as vectors
// (store4's per-element loads were twelve more serial round trips).
// ROPE: the fused q/k/v product of a decode step, as skinny_kernel's (row groups rg and rg + 4 of one image, RoPE + KV append
// in the epilogue).
struct NoRope {};
template <int WAVES, bool XF32, bool SPLIT, bool DUAL, bool NORM = false, bool ROPE = false>
__global__ __launch_bounds__(WAVES * 64, ROPE ? 2 : 4) void skinny_fp8_kernel(LinArgs a, std::conditional_t<ROPE, SmRopeEpi, NoRope> re) {
    static_assert(!NORM || (XF32 && !SPLIT), "fused RMSNorm: fp32 activations, one rounding");
    static_assert(!ROPE || DUAL, "fused RoPE: the two halves of a head ride the dual accumulators");
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = ROPE ? (int)((blockIdx.x >> 2) * 8 + (blockIdx.x & 3)) : (int)blockIdx.x;
    const int KS = a.KS, KSP = (KS + 1) >> 1;
    const int i = lane & 15, g = lane >> 4;
    const u32x4* wp = (const u32x4*)a.w + (size_t)rg * KSP * 64 + lane;
    const u32x4* wp2 = ROPE ? (const u32x4*)a.w + (size_t)(rg + 4) * KSP * 64 + lane : DUAL ? (const u32x4*)a.w2 + (size_t)rg * KSP * 64 + lane : nullptr;
    const bool valid = i < a.M;
    const char* xrow = (const char*)a.x + (size_t)(valid ? i : 0) * a.ldx * (XF32 ? 4 : 2);
    f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};

    // epilogue operands, first in the vector-memory queue (wave 0 stores)
    const int n0 = rg * 16 + g * 4;
    const bool fast = ROPE || ((a.N & 15) == 0 && a.remap_in == 0 && !a.vt && !a.bias && (((size_t)a.wscale | (size_t)a.wscale2) & 15) == 0 &&
                               (!a.residual || ((a.ldr & 3) == 0 && ((size_t)a.residual & 15) == 0)) &&
                               (!a.out_f32 || ((a.ldo & 3) == 0 && ((size_t)a.out_f32 & 15) == 0)) &&
                               (!a.out_bf16 || ((a.ldo_bf16 & 3) == 0 && ((size_t)a.out_bf16 & 7) == 0)));
    f32x4 sc = {1, 1, 1, 1}, sc2 = {1, 1, 1, 1}, res = {0, 0, 0, 0};
    if (fast && wave == 0) {
        sc = *(const f32x4*)(a.wscale + n0);
        if (ROPE) sc2 = *(const f32x4*)(a.wscale + n0 + 64);
        else if (DUAL) sc2 = *(const f32x4*)(a.wscale2 + n0);
        if (!ROPE && a.residual && valid) res = *(const f32x4*)(a.residual + (size_t)i * a.ldr + n0);
    }

    // chunks (16 B per lane and matrix) in flight per wave; 128 VGPRs = 16 waves per CU.  Measured on one box (Mistral-7B decode,
    // tokens/s): 3 or 4 chunks for the dual kernel (138 / 146 VGPRs, one block per CU) 479 / 482 vs 512; 16 waves for the
    // fused q/k/v kernel 511 vs 512
    constexpr int D = (DUAL || (XF32 && !NORM)) ? 2 : 4;
    const int nw = wave < KSP ? (KSP - wave + WAVES - 1) / WAVES : 0;       // chunks of this wave
    bf16_t* const xs = (bf16_t*)(red + (size_t)WAVES * (DUAL ? 8 : 4) * 64 + WAVES * 16);     // NORM: normalised bf16 rows [M][K]
    // raw x of one chunk (two k-steps), fetched with the weights; NORM reads its fragments from LDS when the chunk is consumed
    struct XRaw { std::conditional_t<XF32, f32x4, u32x4> v[XF32 ? 4 : 2]; };
    auto issue = [&](int c, u32x4& q, u32x4& q2, XRaw& x) {
        const int kp = wave + c * WAVES;
        q = __builtin_nontemporal_load(wp + (size_t)kp * 64);
        if (DUAL) q2 = __builtin_nontemporal_load(wp2 + (size_t)kp * 64);
        if constexpr (!NORM) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ks = min(2 * kp + h, KS - 1);              // odd KS: the pair's second half re-reads the first (zeroed below)
                if constexpr (XF32) {
                    x.v[2 * h] = *(const f32x4*)(xrow + (size_t)(ks * 32 + g * 8) * 4);
                    x.v[2 * h + 1] = *(const f32x4*)(xrow + (size_t)(ks * 32 + g * 8) * 4 + 16);
                } else {
                    x.v[h] = *(const u32x4*)(xrow + (size_t)(ks * 32 + g * 8) * 2);
                }
            }
        }
    };
    auto consume = [&](const u32x4& q, const u32x4& q2, const XRaw& x, int c) {
        const int kp = wave + c * WAVES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ks = 2 * kp + h;
            const bool on = valid && ks < KS;
            bf16x8 xh, xl;
            if constexpr (NORM) {
                union { bf16x8 v; u32x4 u; } r;
                r.u = u32x4{0, 0, 0, 0};
                if (on) r.u = *(const u32x4*)(xs + (size_t)i * a.K + ks * 32 + g * 8);
                xh = r.v;
            } else if constexpr (XF32) {
                const f32x4 z = {0, 0, 0, 0};
                split_x<SPLIT>(on ? x.v[2 * h] : z, on ? x.v[2 * h + 1] : z, xh, xl);
            } else {
                union { bf16x8 v; u32x4 u; } r;
                r.u = on ? x.v[h] : u32x4{0, 0, 0, 0};
                xh = r.v;
            }
            const bf16x8 wf = fp8x8_to_bf16(h ? q[2] : q[0], h ? q[3] : q[1]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xh, acc, 0, 0, 0);
            if (SPLIT) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xl, acc, 0, 0, 0);
            if (DUAL) {
                const bf16x8 wf2 = fp8x8_to_bf16(h ? q2[2] : q2[0], h ? q2[3] : q2[1]);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2, xh, acc2, 0, 0, 0);
                if (SPLIT) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2, xl, acc2, 0, 0, 0);
            }
        }
    };
    u32x4 q[D], q2[D];
    XRaw xq[D];
    f32x4 t[4], gm[4];
    if constexpr (NORM) norm_issue<WAVES>(a, wave, lane, t, gm);          // before the weights: vmcnt retires in order
    int done = 0, issued, live;
    if (nw >= 2 * D) {
        // the ring proper.  Its own straight-line entry: with the predicated issues of the short case in front of the loop the
        // compiler's wait-count analysis merges both entries and drains the queue (vmcnt(0)) at the top of every round
#pragma unroll
        for (int u = 0; u < D; ++u) issue(u, q[u], q2[u], xq[u]);
        if constexpr (NORM) norm_finish<WAVES>(a, red + (size_t)WAVES * (DUAL ? 8 : 4) * 64, xs, wave, lane, t, gm);
        issued = live = D;
        for (; issued + D <= nw; issued += D, done += D) {             // steady state: no branches, counted waits
#pragma unroll
            for (int u = 0; u < D; ++u) {
                consume(q[u], q2[u], xq[u], done + u);
                issue(issued + u, q[u], q2[u], xq[u]);
                __builtin_amdgcn_sched_barrier(0);          // or the scheduler sinks every issue behind the last consume: a batch again
            }
        }
    } else {
        issued = live = min(nw, D);
#pragma unroll
        for (int u = 0; u < D; ++u)
            if (u < live) issue(u, q[u], q2[u], xq[u]);
        if constexpr (NORM) norm_finish<WAVES>(a, red + (size_t)WAVES * (DUAL ? 8 : 4) * 64, xs, wave, lane, t, gm);
    }
    const int rem = nw - issued;              // < D chunks left to issue: each takes the registers of the chunk just consumed
#pragma unroll
    for (int u = 0; u < D; ++u)
        if (u < live) {
            consume(q[u], q2[u], xq[u], done + u);
            if (u < rem) issue(issued + u, q[u], q2[u], xq[u]);
        }
#pragma unroll
    for (int u = 0; u < D - 1; ++u)
        if (u < rem) consume(q[u], q2[u], xq[u], issued + u);
    if (WAVES > 1) {
        constexpr int PER = DUAL ? 8 : 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[(wave * PER + r) * 64 + lane] = acc[r];
            if (DUAL) red[(wave * PER + 4 + r) * 64 + lane] = acc2[r];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f, s2 = 0.f;
            for (int w = 0; w < WAVES; ++w) {
                s += red[(w * PER + r) * 64 + lane];
                if (DUAL) s2 += red[(w * PER + 4 + r) * 64 + lane];
            }
            acc[r] = s;
            if (DUAL) acc2[r] = s2;
        }
    }
    if constexpr (ROPE) {
        for (int row = 0; row < a.M; ++row)
            if (i == row) rope_store<false>(re, row, (int)(blockIdx.x >> 2), (int)(blockIdx.x & 3) * 16 + g * 4, acc * sc, acc2 * sc2);
        return;
    } else {
        if (!fast) { store4(a, i, n0, acc, DUAL ? &acc2 : nullptr); return; }
        if (!valid || n0 >= a.N) return;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[r] * sc[r];
            if (DUAL) v = siluf_(v) * (acc2[r] * sc2[r]);
            else v = apply_act_rt(v, a.act);
            o[r] = v + res[r];
        }
        if (a.out_f32) *(f32x4*)(a.out_f32 + (size_t)i * a.ldo + n0) = o;
        if (a.out_bf16) *(u32x2*)(a.out_bf16 + (size_t)i * a.ldo_bf16 + n0) = u32x2{pack16_rt(o[0], o[1], a.f16), pack16_rt(o[2], o[3], a.f16)};
    }
}

#ifndef SKINNY_RING
#define SKINNY_RING 1          // 0: the batch loops of rounds 1-2 for every shape (same-box A/B with tools/build_variant.sh)
#endif
// ------------------------------------------------------------------------------------------------ skinny (M <= 16)
// One block = one 16-row group of W (two groups, one per matrix, when DUAL); its WAVES waves split K (k-step
// ks goes to wave ks % WAVES so the block walks the packed row-group contiguously, 1 KiB per wave-load) and
// reduce through LDS.  Weights stream HBM -> VGPR with non-temporal 16-byte loads; x comes from L2.
// NORM: RMSNorm of the activations fused in front (norm_issue / norm_finish above; decode q/k/v, gate/up, lm_head at one row)
// ROPE: the fused q/k/v product of a decode step (head_dim 128).  Block b streams row groups rg = (b / 4) * 8 + b % 4 and
// rg + 4 of the SAME weight image (the DUAL machinery: two accumulators over one activation fragment), i.e. dims d and d + 64
// of one head land in the same lane, and the epilogue applies rotate_half RoPE and writes q / the K cache / the V^T cache
// directly (rope_store).
template <int WAVES, bool XF32, bool SPLIT, bool DUAL, int MB, bool NORM = false, bool ROPE = false, bool F16 = false>   // MB: 16-row activation blocks (M <= 16 * MB); F16: IEEE fp16 operands
__global__ __launch_bounds__(WAVES * 64) void skinny_kernel(LinArgs a, std::conditional_t<ROPE, SmRopeEpi, NoRope> re) {
    static_assert(!NORM || (XF32 && !SPLIT && MB == 1), "fused RMSNorm: fp32 activations, one rounding, <= 16 rows");
    static_assert(!ROPE || (DUAL && MB == 1), "fused RoPE: the two halves of a head ride the dual accumulators");
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = ROPE ? (int)((blockIdx.x >> 2) * 8 + (blockIdx.x & 3)) : (int)blockIdx.x;
    const int KS = a.KS;
    const int i = lane & 15, g = lane >> 4;
    const bf16x8* wp = a.w + (size_t)rg * KS * 64 + lane;
    const bf16x8* wp2 = ROPE ? a.w + (size_t)(rg + 4) * KS * 64 + lane : DUAL ? a.w2 + (size_t)rg * KS * 64 + lane : nullptr;
    bool valid[MB];
    const char* xrow[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        valid[mb] = mb * 16 + i < a.M;
        xrow[mb] = (const char*)a.x + (size_t)(valid[mb] ? mb * 16 + i : 0) * a.ldx * (XF32 ? 4 : 2);
    }
    f32x4 acc[MB], acc2[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) { acc[mb] = f32x4{0, 0, 0, 0}; acc2[mb] = f32x4{0, 0, 0, 0}; }

    // residual rows of the epilogue, fetched now (wave 0 stores): at the end of the kernel the load was a bare L2 round trip
    // on the critical path of every o / down projection of the decode step
    const bool pre_res = MB == 1 && !DUAL && a.residual && !a.bias && a.act == SM_ACT_NONE && !a.wscale && a.remap_in == 0 &&
                         (a.N & 3) == 0 && (a.ldr & 3) == 0;
    f32x4 res0 = {0, 0, 0, 0};
    if (pre_res && wave == 0 && i < a.M) res0 = *(const f32x4*)(a.residual + (size_t)i * a.ldr + rg * 16 + g * 4);
    constexpr int U = MB == 1 ? 4 : 2;
    int ks = wave;
    // NORM: normalised bf16 rows [M][K] in LDS behind the reduction scratch
    bf16_t* const xs = (bf16_t*)(red + (size_t)WAVES * (DUAL ? 8 : 4) * 64 + WAVES * 16);
    // x fragment of row i at k-step kk (NORM: normalised on the fly)
    auto xfrag = [&](int mb, int kk, bf16x8& xh, bf16x8& xl) {
        if constexpr (NORM) {
            union { bf16x8 v; u32x4 u; } r;
            r.u = u32x4{0, 0, 0, 0};
            if (valid[0]) r.u = *(const u32x4*)(xs + (size_t)i * a.K + kk * 32 + g * 8);
            xh = r.v;
        } else {
            load_x<XF32, SPLIT, F16>(xrow[mb], xcol(a, kk * 32 + g * 8), valid[mb], xh, xl);
        }
    };
    auto mfmas = [&](const bf16x8 (&wa)[U], const bf16x8 (&wb)[U], int k0) {
        if constexpr (NORM) {
            // two k-steps' x / gamma fragments at a time: all four at once cost 32 more VGPRs and a block per CU
#pragma unroll
            for (int u0 = 0; u0 < U; u0 += 2) {
                bf16x8 xh[2], xl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) xfrag(0, k0 + (u0 + u) * WAVES, xh[u], xl[u]);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[0] = mfma16<F16>(wa[u0 + u], xh[u], acc[0]);
                    if (DUAL) acc2[0] = mfma16<F16>(wb[u0 + u], xh[u], acc2[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            bf16x8 xh[U], xl[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xfrag(mb, k0 + u * WAVES, xh[u], xl[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[mb] = mfma16<F16>(wa[u], xh[u], acc[mb]);
                if (SPLIT) acc[mb] = mfma16<F16>(wa[u], xl[u], acc[mb]);
                if (DUAL) {
                    acc2[mb] = mfma16<F16>(wb[u], xh[u], acc2[mb]);
                    if (SPLIT) acc2[mb] = mfma16<F16>(wb[u], xl[u], acc2[mb]);
                }
            }
        }
    };
#if SKINNY_RING
    if constexpr (MB == 1 && !SPLIT) {
        // <= 16 rows, one rounding of x: the register ring of skinny_fp8_kernel (D k-steps of 1 KiB per matrix in flight per
        // wave, x fragments fetched with the weights, same accumulation order as the batch loops below: bit-identical results)
        constexpr int D = 4;          // deeper rings for the single-matrix kernels (o / down) measured slower: D = 6 / 8 -> 336 / 340 vs 345 tokens/s
        const int nw = wave < KS ? (KS - wave + WAVES - 1) / WAVES : 0;
        struct XRaw { std::conditional_t<XF32, f32x4, u32x4> v[XF32 ? 2 : 1]; };
        auto issue = [&](int c, bf16x8& q, bf16x8& q2, XRaw& x) {
            const int kk = wave + c * WAVES;
            q = __builtin_nontemporal_load(wp + (size_t)kk * 64);
            if (DUAL) q2 = __builtin_nontemporal_load(wp2 + (size_t)kk * 64);
            if constexpr (!NORM) {
                if constexpr (XF32) {
                    const int kc = xcol(a, kk * 32 + g * 8);
                    x.v[0] = *(const f32x4*)(xrow[0] + (size_t)kc * 4);
                    x.v[1] = *(const f32x4*)(xrow[0] + (size_t)kc * 4 + 16);
                } else {
                    x.v[0] = *(const u32x4*)(xrow[0] + (size_t)(kk * 32 + g * 8) * 2);
                }
            }
        };
        auto consume = [&](const bf16x8& q, const bf16x8& q2, const XRaw& x, int c) {
            bf16x8 xh, xl;
            if constexpr (NORM) {
                xfrag(0, wave + c * WAVES, xh, xl);
            } else if constexpr (XF32) {
                const f32x4 z = {0, 0, 0, 0};
                split_x<false, F16>(valid[0] ? x.v[0] : z, valid[0] ? x.v[1] : z, xh, xl);
            } else {
                union { bf16x8 v; u32x4 u; } r;
                r.u = valid[0] ? x.v[0] : u32x4{0, 0, 0, 0};
                xh = r.v;
            }
            acc[0] = mfma16<F16>(q, xh, acc[0]);
            if (DUAL) acc2[0] = mfma16<F16>(q2, xh, acc2[0]);
        };
        bf16x8 q[D], q2[D];
        XRaw xq[D];
        f32x4 t[4], gm[4];
        float* nsum = red + (size_t)WAVES * (DUAL ? 8 : 4) * 64;
        if constexpr (NORM) norm_issue<WAVES>(a, wave, lane, t, gm);
        int done = 0, issued, live;
        if (nw >= 2 * D) {
#pragma unroll
            for (int u = 0; u < D; ++u) issue(u, q[u], q2[u], xq[u]);
            if constexpr (NORM) norm_finish<WAVES, F16>(a, nsum, xs, wave, lane, t, gm);
            issued = live = D;
            for (; issued + D <= nw; issued += D, done += D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    consume(q[u], q2[u], xq[u], done + u);
                    issue(issued + u, q[u], q2[u], xq[u]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            issued = live = min(nw, D);
#pragma unroll
            for (int u = 0; u < D; ++u)
                if (u < live) issue(u, q[u], q2[u], xq[u]);
            if constexpr (NORM) norm_finish<WAVES, F16>(a, nsum, xs, wave, lane, t, gm);
        }
        const int rem = nw - issued;
#pragma unroll
        for (int u = 0; u < D; ++u)
            if (u < live) {
                consume(q[u], q2[u], xq[u], done + u);
                if (u < rem) issue(issued + u, q[u], q2[u], xq[u]);
            }
#pragma unroll
        for (int u = 0; u < D - 1; ++u)
            if (u < rem) consume(q[u], q2[u], xq[u], issued + u);
    } else
#endif
    {
        if constexpr (NORM) {
            float* nsum = red + (size_t)WAVES * (DUAL ? 8 : 4) * 64;
            f32x4 t[4], gm[4];
            norm_issue<WAVES>(a, wave, lane, t, gm);
            bf16x8 wa0[U], wb0[U];
            const bool have0 = ks + (U - 1) * WAVES < KS;
            if (have0) {
    #pragma unroll
                for (int u = 0; u < U; ++u) {
                    wa0[u] = __builtin_nontemporal_load(wp + (size_t)(ks + u * WAVES) * 64);
                    if (DUAL) wb0[u] = __builtin_nontemporal_load(wp2 + (size_t)(ks + u * WAVES) * 64);
                }
            }
            norm_finish<WAVES, F16>(a, nsum, xs, wave, lane, t, gm);
            if (have0) { mfmas(wa0, wb0, ks); ks += U * WAVES; }
        }
        for (; ks + (U - 1) * WAVES < KS; ks += U * WAVES) {
            bf16x8 wa[U], wb[U];
    #pragma unroll
            for (int u = 0; u < U; ++u) {
                wa[u] = __builtin_nontemporal_load(wp + (size_t)(ks + u * WAVES) * 64);
                if (DUAL) wb[u] = __builtin_nontemporal_load(wp2 + (size_t)(ks + u * WAVES) * 64);
            }
            if constexpr (NORM) { mfmas(wa, wb, ks); continue; }
    #pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                bf16x8 xh[U], xl[U];
    #pragma unroll
                for (int u = 0; u < U; ++u) load_x<XF32, SPLIT, F16>(xrow[mb], xcol(a, (ks + u * WAVES) * 32 + g * 8), valid[mb], xh[u], xl[u]);
    #pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc[mb] = mfma16<F16>(wa[u], xh[u], acc[mb]);
                    if (SPLIT) acc[mb] = mfma16<F16>(wa[u], xl[u], acc[mb]);
                    if (DUAL) {
                        acc2[mb] = mfma16<F16>(wb[u], xh[u], acc2[mb]);
                        if (SPLIT) acc2[mb] = mfma16<F16>(wb[u], xl[u], acc2[mb]);
                    }
                }
            }
        }
        for (; ks < KS; ks += WAVES) {
            bf16x8 wa = __builtin_nontemporal_load(wp + (size_t)ks * 64), wb;
            if (DUAL) wb = __builtin_nontemporal_load(wp2 + (size_t)ks * 64);
    #pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                bf16x8 xh, xl;
                xfrag(mb, ks, xh, xl);
                acc[mb] = mfma16<F16>(wa, xh, acc[mb]);
                if (SPLIT) acc[mb] = mfma16<F16>(wa, xl, acc[mb]);
                if (DUAL) {
                    acc2[mb] = mfma16<F16>(wb, xh, acc2[mb]);
                    if (SPLIT) acc2[mb] = mfma16<F16>(wb, xl, acc2[mb]);
                }
            }
        }
    }
    if (WAVES > 1) {
        // cross-wave K reduction in a FIXED order (deterministic): red[wave][mb][reg][lane]
        constexpr int PER = (DUAL ? 8 : 4) * MB;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[(wave * PER + mb * (DUAL ? 8 : 4) + r) * 64 + lane] = acc[mb][r];
                if (DUAL) red[(wave * PER + mb * 8 + 4 + r) * 64 + lane] = acc2[mb][r];
            }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f, s2 = 0.f;
                for (int w = 0; w < WAVES; ++w) {
                    s += red[(w * PER + mb * (DUAL ? 8 : 4) + r) * 64 + lane];
                    if (DUAL) s2 += red[(w * PER + mb * 8 + 4 + r) * 64 + lane];
                }
                acc[mb][r] = s;
                if (DUAL) acc2[mb][r] = s2;
            }
    }
    if constexpr (ROPE) {
        // rows walked with a wave-uniform index: the per-stream position / cache pointers are scalar loads from the argument block
        for (int row = 0; row < a.M; ++row)
            if (i == row) rope_store<F16>(re, row, (int)(blockIdx.x >> 2), (int)(blockIdx.x & 3) * 16 + g * 4, acc[0], acc2[0]);
        return;
    }
    if (pre_res) {
        LinArgs b = a;
        b.residual = nullptr;
        store4(b, i, rg * 16 + g * 4, acc[0] + res0, nullptr);
        return;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) store4(a, mb * 16 + i, rg * 16 + g * 4, acc[mb], DUAL ? &acc2[mb] : nullptr);
}

// ------------------------------------------------------------------------------------------------ tiled GEMM
// 128(m) x 128(n) output tile, BK = 64, 4 waves as 2(n) x 2(m), each wave 4x4 fragments of 16x16.
// Operands reach LDS by global_load_lds (16 B/lane, 1 KiB per wave-instruction):
//   W tile  : 16 packed (row-group, k-step) chunks of 1 KiB -> LDS image already in fragment order
//   X tile  : [128][64] bf16, 128-byte rows, 16-byte chunk index XOR-swizzled with (row & 7); the swizzle is
//             applied on the per-lane SOURCE address (the LDS destination of global_load_lds is lane-linear)
//             and again on the ds_read_b128 address.
// Two LDS stages (64 KiB) -> 2 blocks/CU; loads of tile t+1 are in flight while tile t is multiplied.
#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_STAGE_BYTES 32768

// Fast whole-row epilogue of one [128 m][128 n] fp32 tile staged in LDS.  The output / residual / bias pointers are
// __restrict__ FUNCTION PARAMETERS on purpose: without the no-alias guarantee hipcc orders every pass's (possibly
// aliasing, in-place) residual load behind the previous pass's store with s_waitcnt vmcnt(0), which serialises the
// passes on the store round trip (measured: 25 us of a 34 us QKV GEMM).  In-place residual is still safe: the
// same lane reads an element before it writes it, and different passes touch different rows.  ACT is a
// compile-time activation so the 64 inlined element epilogues stay small (I-cache).
template <int ACT, int THREADS, bool F16>
__device__ __forceinline__ void tile_rows_epilogue(const char* __restrict__ smem, const float* __restrict__ bias,
                                                   const float* __restrict__ residual, int ldr, float* __restrict__ out_f32,
                                                   int ldo, bf16_t* __restrict__ out_bf16, int ldob, int m0, int n0,
                                                   int M, int tid) {
    const int chunk = tid & 31;
    const int n = n0 + chunk * 4;
    f32x4 b4 = {0, 0, 0, 0};
    if (bias) b4 = *(const f32x4*)(bias + n);
    constexpr int RPP = THREADS / 32;          // rows per pass
#pragma unroll
    for (int p0 = 0; p0 < 128 / RPP; p0 += 4) {
        f32x4 v[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ml = (p0 + u) * RPP + (tid >> 5);
            v[u] = *(const f32x4*)(smem + ml * 512 + ((chunk ^ (ml & 31)) * 16));
            r[u] = f32x4{0, 0, 0, 0};
            if (residual && m0 + ml < M) r[u] = *(const f32x4*)(residual + (size_t)(m0 + ml) * ldr + n);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + (p0 + u) * RPP + (tid >> 5);
            if (m >= M) continue;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[u][j] + b4[j];
                if (ACT == SM_ACT_QUICK_GELU) t = t * sigmoidf_(1.702f * t);
                o[j] = t + r[u][j];
            }
            if (out_f32) *(f32x4*)(out_f32 + (size_t)m * ldo + n) = o;
            if (out_bf16) *(u32x2*)(out_bf16 + (size_t)m * ldob + n) = u32x2{pack16<F16>(o[0], o[1]), pack16<F16>(o[2], o[3])};
        }
    }
}

// WV = 8: the same tile on 8 waves (4(n) x 2(m), each 2x4 fragments): with one block per CU a 4-wave block leaves ONE wave
// per SIMD, whose DMA issues, fragment reads and MFMAs then run strictly one after the other (~0.75 us per k-tile against
// 0.27 us of MFMA issue); two waves per SIMD overlap each other.
// STAGES = 4 (8 waves only): a RING of four 32-KiB stages, loads issued THREE k-tiles ahead, one barrier per k-tile.  The two-stage
// loop issues the loads of tile t+1 only after the trailing barrier of tile t-1, so every k-tile pays one full load latency
// (L2 ~0.2 us, HBM ~0.5 us: measured 0.6-0.75 us per k-tile where the 32 MFMAs per SIMD take 0.22); a CU can stage 145 KB/us
// from L2 (tools/experiments/stage_ubench.hip), i.e. a 32-KiB stage in 0.22 us -- with three stages in flight the loop is
// MFMA-bound.  One frame through the ViT (M = 577, 40-160 tiles, one partial round) is the case this is for.
#ifdef SM_GEMM128_TIMELINE
__device__ long long* g_tl128;
extern "C" int sm_debug_set_timeline128(long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_tl128), &p, sizeof(p)) == hipSuccess ? 0 : -2; }
#define TL128(k) do { if (threadIdx.x == 0 && g_tl128) g_tl128[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TL128(k) do {} while (0)
#endif
template <int ACT, int WV = 4, bool F16 = false, int STAGES = 2>   // ACT: compile-time activation of the fast epilogue (SM_ACT_NONE / SM_ACT_QUICK_GELU); -1: runtime a.act
__global__ __launch_bounds__(WV * 64, WV == 4 ? 2 : 1) void gemm_kernel(LinArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TL128(0);
    constexpr int NF = 16 / WV;                // 16-row weight fragments per wave: 4 (2 x 2 waves) or 2 (4 x 2 waves)
    constexpr int J = 16 / WV;                 // W pieces (and X pieces) of 1 KiB each wave stages per k-tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int wn = wave >> 1, wm = wave & 1;

    // XCD-aware tile order: block b runs on XCD b % 8; give every XCD a contiguous band of tile ids (bijective
    // for any grid size) and walk n fastest inside it so the X band stays in that XCD's L2.
    const int nblk = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // (a.n_band, round 6: with fewer row tiles than column tiles -- a decode step of 129..512 streams, a short prefill chunk -- W is the big operand and every XCD that holds a
    // different row tile of the same columns fetched it again: 4 x 117 MB for down_proj at 512 rows, which is what that launch took at HBM speed.  m fastest inside the band puts
    // the row tiles of one W tile side by side on one XCD)
    int tile_m, tile_n;
    if (a.n_band) { tile_n = bid / tiles_m; tile_m = bid - tile_n * tiles_m; }
    else { tile_m = bid / tiles_n; tile_n = bid - tile_m * tiles_n; }

    // split-K (prefill-sized M, few tiles): blockIdx.y owns k-tiles [kt0, KT) of the K loop and writes its raw fp32 partial tile
    // to the workspace slab a.out_f32 + blockIdx.y * M * ldo (the launcher redirected the outputs; splitk_reduce_kernel sums the
    // slabs in a fixed order and applies the real epilogue)
    const int KTall = a.KS >> 1;
    const int kt0 = (int)((long)KTall * blockIdx.y / gridDim.y), KT = (int)((long)KTall * (blockIdx.y + 1) / gridDim.y);
    if (gridDim.y > 1) a.out_f32 += (size_t)blockIdx.y * a.M * a.ldo;
    const char* xbase = (const char*)a.x;

    // per-lane source addresses for the 4 + 4 staging loads of this wave
    const char* wsrc[J];
    const char* xsrc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        int c = wave * J + j;                 // W chunk: rg_local = c >> 1, ks_local = c & 1
        int rgg = tile_n * 8 + (c >> 1);
        if (rgg >= a.NRG) rgg = a.NRG - 1;
        wsrc[j] = (const char*)a.w + ((size_t)rgg * a.KS + (c & 1)) * 1024 + lane * 16;
        int row = c * 8 + (lane >> 3);        // X piece: 8 rows x 128 B
        int mg = tile_m * GEMM_BM + row;
        if (mg >= a.M) mg = a.M - 1;
        int chunk = (lane & 7) ^ (row & 7);
        xsrc[j] = xbase + ((size_t)mg * a.ldx + chunk * 8) * 2;
    }
    f32x4 acc[NF][4];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) acc[nf][mf] = f32x4{0, 0, 0, 0};

    auto stage = [&](int kt, int buf) {
        char* sb = smem + buf * GEMM_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            int c = wave * J + j;
            glds16(wsrc[j] + (size_t)kt * 2048, sb + c * 1024);
            glds16(xsrc[j] + (size_t)kt * 128, sb + 16384 + c * 1024);
        }
    };

    if constexpr (STAGES >= 4) {
        static_assert(WV == 8, "the ring loop counts 4 loads per wave and stage");
        // ---- ring loop.  The two-stage loop below puts every wave of the block through the same three phases between two barriers
        // (issue DMA | read fragments | multiply): with the waves in lockstep the CU's vector-memory path, LDS and matrix pipes are
        // busy one after the other -- 0.83 us per k-tile measured with everything L2-warm, where each resource alone needs <= 0.22.
        // Here a wave keeps TWO fragment sets in registers and the iteration is skewed around its single barrier:
        //     read set B <- (stage t, k 32..63)      | multiply set A (stage t, k 0..31)
        //     wait + barrier: stage t+1 has landed, nobody reads stage t-1 any more
        //     issue DMA of stage t+STAGES-1 (into the slot of t-1), read set A <- (stage t+1, k 0..31)   | multiply set B
        // so the fragment reads and the DMA issue of every wave sit under its own MFMAs.  Past the end the last tile is fetched
        // again (never read) and a stale slot is read (never multiplied into a stored result): the body has no branches.
        constexpr int AHEAD = STAGES - 1;
        const int n_it = KT - kt0;
#pragma unroll
        for (int p = 0; p < AHEAD; ++p) stage(kt0 + p < KT ? kt0 + p : KT - 1, p);
        const int wbase = wn * NF * 2048 + lane * 16;
        int xbase[2];
#pragma unroll
        for (int ksl = 0; ksl < 2; ++ksl) xbase[ksl] = 16384 + (wm * 64 + i) * 128 + (((ksl * 4 + g) ^ (i & 7)) * 16);
        bf16x8 wA[NF], xA[4], wB[NF], xB[4];
        auto ldfrag = [&](const char* sb, int ksl, bf16x8 (&wf)[NF], bf16x8 (&xf)[4]) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) wf[nf] = *(const bf16x8*)(sb + wbase + nf * 2048 + ksl * 1024);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) xf[mf] = *(const bf16x8*)(sb + xbase[ksl] + mf * 2048);
        };
        auto mma = [&](const bf16x8 (&wf)[NF], const bf16x8 (&xf)[4]) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) acc[nf][mf] = mfma16<F16>(wf[nf], xf[mf], acc[nf][mf]);
        };
        if constexpr (AHEAD == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TL128(1);
        ldfrag(smem, 0, wA, xA);
        int slot = 0;
        for (int it = 0; it < n_it; ++it) {
            const char* sb = smem + slot * GEMM_STAGE_BYTES;
            ldfrag(sb, 1, wB, xB);
            mma(wA, xA);
            // set A was read an iteration ago: its MFMAs start at once and set B's reads go into their gaps (reads in front would
            // put set B under the same lgkmcnt(0) as set A: the counter is in order and hipcc cannot count across the back edge)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
            if constexpr (AHEAD == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int nslot = slot + 1 == STAGES ? 0 : slot + 1;
            const int fslot = slot == 0 ? STAGES - 1 : slot - 1;          // (it + STAGES - 1) % STAGES
            const int ktn = kt0 + it + AHEAD;
            stage(ktn < KT ? ktn : KT - 1, fslot);
            ldfrag(smem + nslot * GEMM_STAGE_BYTES, 0, wA, xA);
            mma(wB, xB);
            // one MFMA in front, then DMA issues and fragment reads in the gaps of the remaining MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            slot = nslot;
        }
        // the dummy stages still in flight land in the slots the epilogue stages the tile through; the stale fragment reads must be done
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
    stage(kt0, 0);
    for (int kt = kt0; kt < KT; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < KT) {
            stage(kt + 1, buf ^ 1);
            if (J == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const char* sw = smem + buf * GEMM_STAGE_BYTES;
        const char* sx = sw + 16384;
#ifdef SM_GEMM128_TIMELINE
        if (kt == kt0) TL128(1);
#endif
#pragma unroll
        for (int ksl = 0; ksl < 2; ++ksl) {
            bf16x8 wf[NF], xf[4];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                wf[nf] = *(const bf16x8*)(sw + (((wn * NF + nf) * 2 + ksl) * 1024) + lane * 16);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                int ml = wm * 64 + mf * 16 + i;
                xf[mf] = *(const bf16x8*)(sx + ml * 128 + (((ksl * 4 + g) ^ (ml & 7)) * 16));
            }
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
                    acc[nf][mf] = mfma16<F16>(wf[nf], xf[mf], acc[nf][mf]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    }
    TL128(2);
    // ---- epilogue: stage the fp32 tile through LDS ([128 m][128 n] fp32 = the whole 64 KiB; 16-byte chunk index
    // XOR-swizzled with (m & 31) so both the fragment-shaped writes and the row-shaped reads are conflict-free),
    // then every wave walks whole rows: 512 B contiguous per row to HBM.
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            int ml = wm * 64 + mf * 16 + i;
            int chunk = (wn * NF + nf) * 4 + g;
            *(f32x4*)(smem + ml * 512 + ((chunk ^ (ml & 31)) * 16)) = acc[nf][mf];
        }
    __syncthreads();
    if (a.vt && tile_n * GEMM_BN >= a.vt_n0) {
        // V^T side output: lanes walk m (= s, the contiguous dim of vt), one n per wave per pass
        const int nh = (a.N - a.vt_n0) / a.vt_dh;
        for (int pass = 0; pass < 128 / WV; ++pass) {
            const int nl = pass * WV + wave;
            const int n = tile_n * GEMM_BN + nl;
            if (n >= a.N) continue;
            const int c = n - a.vt_n0;
            const int h = c / a.vt_dh, d = c - h * a.vt_dh;
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int ml = lane + 64 * h2;
                const int m = tile_m * GEMM_BM + ml;
                if (m < a.M) {
                    float v = *(const float*)(smem + ml * 512 + (((nl >> 2) ^ (ml & 31)) * 16) + (nl & 3) * 4) + bv;
                    const int b = m / a.vt_S, sidx = m - b * a.vt_S;
                    a.vt[((size_t)(b * nh + h) * a.vt_dh + d) * a.vt_ld + sidx] = (bf16_t)cvt16<F16>(apply_act_rt(v, a.act));
                }
            }
        }
    } else if (ACT >= 0 && a.remap_in == 0 && (tile_n + 1) * GEMM_BN <= a.N && (a.ldo & 3) == 0 && (a.ldo_bf16 & 3) == 0 &&
               (a.ldr & 3) == 0) {
        tile_rows_epilogue<ACT, WV * 64, F16>(smem, a.bias, a.residual, a.ldr, a.out_f32, a.ldo, a.out_bf16, a.ldo_bf16,
                                tile_m * GEMM_BM, tile_n * GEMM_BN, a.M, tid);
    } else {
        for (int pass = 0; pass < 128 / (WV * 2); ++pass) {
            const int ml = pass * (WV * 2) + (tid >> 5);
            const int chunk = tid & 31;
            f32x4 v = *(const f32x4*)(smem + ml * 512 + ((chunk ^ (ml & 31)) * 16));
            store4(a, tile_m * GEMM_BM + ml, tile_n * GEMM_BN + chunk * 4, v, nullptr);
        }
    }
    TL128(3);
}

// sum of the split-K slabs (fixed order: deterministic) + the real epilogue; one thread per 4 outputs.  ws2 != nullptr: the
// slabs of the second (dual / SwiGLU) weight, combined by store4 as silu(v) * v2.
__global__ void splitk_reduce_kernel(LinArgs a, const float* __restrict__ ws, const float* __restrict__ ws2, int S, int ldw) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n4 = (a.N + 3) >> 2;
    if (t >= (size_t)a.M * n4) return;
    const int m = (int)(t / n4), n0 = (int)(t % n4) * 4;
    f32x4 v = {0, 0, 0, 0}, v2 = {0, 0, 0, 0};
    const size_t sstride = (size_t)a.M * ldw, o0 = (size_t)m * ldw + n0;
    if (n0 + 3 < a.N) {
        // slabs fetched 8 at a time (independent loads in flight together), summed in slab order
        for (int s0 = 0; s0 < S; s0 += 8) {
            f32x4 t[8], t2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                t[u] = s0 + u < S ? *(const f32x4*)(ws + (size_t)(s0 + u) * sstride + o0) : f32x4{0, 0, 0, 0};
                if (ws2) t2[u] = s0 + u < S ? *(const f32x4*)(ws2 + (size_t)(s0 + u) * sstride + o0) : f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < S) { v += t[u]; if (ws2) v2 += t2[u]; }
        }
    } else {
        for (int s = 0; s < S; ++s)
            for (int r = 0; r < 4; ++r)
                if (n0 + r < a.N) { v[r] += ws[(size_t)s * sstride + o0 + r]; if (ws2) v2[r] += ws2[(size_t)s * sstride + o0 + r]; }
    }
    store4(a, m, n0, v, ws2 ? &v2 : nullptr);
}

// Split-K slabs -> finished fp32 row AND its LayerNorm in one pass (sm_linear_t.post_ln_*; one frame through the ViT).  One wave per
// row of D = 256 * NV columns, a lane owns columns lane*4 + j*256 -- the layout and the arithmetic of norm_wave_fixed_kernel (mean,
// then the sum of squared deviations, in registers), and the slab sum / bias / residual order of splitk_reduce_kernel + store4: the
// two outputs are bit for bit what the two separate launches wrote.  Replaces two dependent launches (~5 us of fixed cost each at
// this size) and the re-read of the row by one.
template <int NV>
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(LinArgs a, const float* __restrict__ ws, int S, int ldw, PostLn ln) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const size_t sstride = (size_t)a.M * ldw;
    const float* wr = ws + (size_t)row * ldw + lane * 4;
    f32x4 t[8][NV], r[NV], b[NV], v[NV];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < NV; ++j) t[u][j] = u < S ? *(const f32x4*)(wr + (size_t)u * sstride + j * 256) : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + j * 256;
        r[j] = a.residual ? *(const f32x4*)(a.residual + (size_t)row * a.ldr + c) : f32x4{0, 0, 0, 0};
        b[j] = a.bias ? *(const f32x4*)(a.bias + c) : f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < S) acc += t[u][j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float o = acc[e];
            if (a.bias) o += b[j][e];
            if (a.residual) o += r[j][e];
            v[j][e] = o;
        }
        *(f32x4*)(a.out_f32 + (size_t)row * a.ldo + lane * 4 + j * 256) = v[j];
    }
    float mu, rstd;
    ln_row_stats<NV>(v, ln.eps, mu, rstd);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + j * 256;
        const f32x4 gm = *(const f32x4*)(ln.gamma + c), bt = *(const f32x4*)(ln.beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ln_apply(v[j][e], mu, rstd, gm[e], bt[e]);
        *(u32x2*)(ln.out + (size_t)row * ln.ldo + c) = u32x2{pack16_rt(o[0], o[1], a.f16), pack16_rt(o[2], o[3], a.f16)};
    }
}

// The same for the weight-streaming products of 17..32 rows (connector / gate pass; skinny_lds_kernel leaves K-slice slabs): ONE BLOCK
// per row, one thread per 4 columns (N / 4 <= 1024 threads: every slab load of the row is in flight at once), slab sum in slab order +
// store4's epilogue (row scale of fp8 weights, bias, activation, residual) -> the fp32 row, then LayerNorm (beta != NULL: mean, then
// squared deviations) or RMSNorm of the row across the block, optional activation -> fp32 and / or 16-bit.  Replaces
// splitk_reduce_kernel + a norm launch.
__global__ __launch_bounds__(1024) void splitk_reduce_norm_rows_kernel(LinArgs a, const float* __restrict__ ws, int S, PostLn ln) {
    __shared__ float red[16];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const size_t sstride = (size_t)a.M * a.N;
    const int c = tid * 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int s0 = 0; s0 < S; s0 += 8) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = s0 + u < S ? *(const f32x4*)(ws + (size_t)(s0 + u) * sstride + (size_t)m * a.N + c) : f32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < S) acc += t[u];
    }
    f32x4 v;
    // bias / residual / row scales as 16-byte loads (c is a multiple of 4; the residual row needs ldr % 4 == 0, which every caller of this pass
    // guarantees -- checked here so that an odd stride still takes the element loads): at 2048 rows (prefill chunks) the four element loads per
    // operand made this pass 39 us for 151 MB
    // (round-5 advisor: the BASE pointers too -- a caller's sliced bias / scale / residual with 4-byte alignment takes the element loads, not a misaligned-address fault)
    const bool vec = (a.ldr & 3) == 0 && ((((uintptr_t)a.residual) | ((uintptr_t)a.bias) | ((uintptr_t)a.wscale)) & 15) == 0;
    f32x4 r4 = {0, 0, 0, 0}, b4 = {0, 0, 0, 0}, s4 = {1.f, 1.f, 1.f, 1.f};
    if (a.residual && vec) r4 = *(const f32x4*)(a.residual + (size_t)m * a.ldr + c);
    if (a.bias) { if (vec) b4 = *(const f32x4*)(a.bias + c); else b4 = f32x4{a.bias[c], a.bias[c + 1], a.bias[c + 2], a.bias[c + 3]}; }
    if (a.wscale) { if (vec) s4 = *(const f32x4*)(a.wscale + c); else s4 = f32x4{a.wscale[c], a.wscale[c + 1], a.wscale[c + 2], a.wscale[c + 3]}; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = acc[e];
        if (a.wscale) t *= s4[e];
        if (a.bias) t += b4[e];
        if (a.act != SM_ACT_NONE) t = apply_act_rt(t, a.act);           // (a real call: not for the plain case)
        if (a.residual) t += vec ? r4[e] : a.residual[(size_t)m * a.ldr + c + e];
        v[e] = t;
    }
    if (a.out_f32) *(f32x4*)(a.out_f32 + (size_t)m * a.ldo + c) = v;
    const f32x4 gm = *(const f32x4*)(ln.gamma + c);
    f32x4 bt = {0, 0, 0, 0};
    if (ln.beta) bt = *(const f32x4*)(ln.beta + c);
    const float inv_n = 1.0f / (float)a.N;
    auto block_sum = [&](float x) {
        x = wave_sum(x);
        __syncthreads();
        if (lane == 0) red[w] = x;
        __syncthreads();
        float tot = 0.f;
        for (int i = 0; i < nw; ++i) tot += red[i];
        return tot;
    };
    float mu = 0.f, rstd;
    if (ln.beta) {
        mu = block_sum((v[0] + v[1]) + (v[2] + v[3])) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mu; q += d * d; }
        rstd = rsqrtf(block_sum(q) * inv_n + ln.eps);
    } else {
        rstd = rsqrtf(block_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) * inv_n + ln.eps);
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = ln.beta ? (v[e] - mu) * rstd * gm[e] + bt[e] : gm[e] * (v[e] * rstd);
        o[e] = ln.act != SM_ACT_NONE ? apply_act_rt(t, ln.act) : t;
    }
    if (ln.out_f32) *(f32x4*)(ln.out_f32 + (size_t)m * ln.ldo + c) = o;
    if (ln.out) *(u32x2*)(ln.out + (size_t)m * ln.ldo + c) = u32x2{pack16_rt(o[0], o[1], a.f16), pack16_rt(o[2], o[3], a.f16)};
}

// per-HIP-stream split-K workspace (grown on demand; steady state allocates nothing)
#include <map>
#include <mutex>
static std::mutex g_ws_mu;
static std::map<hipStream_t, std::pair<float*, size_t>> g_ws;
static std::map<hipStream_t, std::pair<float*, size_t>> g_dual_ws;      // fp32 gate | up rows of SwiGLU-dual calls outside the 256 x 256 kernel (sm_linear)
int splitk_workspace(hipStream_t st, size_t bytes, float** out) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto& e = g_ws[st];
    if (e.second < bytes) {
        if (e.first) { SM_HIP(hipStreamSynchronize(st)); (void)hipFree(e.first); e.first = nullptr; e.second = 0; }
        SM_HIP(hipMalloc((void**)&e.first, bytes));
        e.second = bytes;
    }
    *out = e.first;
    return SM_OK;
}
static int dual_workspace(hipStream_t st, size_t bytes, float** out);
// Reserve the per-HIP-stream scratch of the tiled products up front (round-5 advisor: the split-K slabs and the unfused SwiGLU rows were grown on demand -- a
// hipStreamSynchronize + hipFree + hipMalloc in the middle of a request, illegal under stream capture and able to run out of memory there instead of at
// set-up).  The model calls this when it creates the LLM workspaces of a HIP stream, with the worst case of a prefill chunk; later calls never grow below that.
int sm_linear_reserve(hipStream_t st, size_t slab_bytes, size_t dual_bytes) {
    float* p = nullptr;
    int rc = slab_bytes ? splitk_workspace(st, slab_bytes, &p) : SM_OK;
    if (!rc && dual_bytes) rc = dual_workspace(st, dual_bytes, &p);
    return rc;
}
int launch_splitk_reduce(const LinArgs& a, const float* ws, int S, int ldw, hipStream_t st) {
    const size_t nthr = (size_t)a.M * ((a.N + 3) / 4);
    splitk_reduce_kernel<<<(unsigned)((nthr + 255) / 256), 256, 0, st>>>(a, ws, nullptr, S, ldw);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// bf16 expansion of fp8 weights for M > 16 (per HIP stream, grown on demand; slot 0 = w, slot 1 = w2)
static std::map<hipStream_t, std::pair<void*, size_t>> g_dq[2];
void release_fp8_workspace(hipStream_t st);                                 // gemm_fp8.hip
// a HIP stream this library created is about to be destroyed (already synchronised): drop the slabs keyed on it
void release_stream_workspaces(hipStream_t st) {
    {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        auto w = g_ws.find(st);
        if (w != g_ws.end()) { if (w->second.first) (void)hipFree(w->second.first); g_ws.erase(w); }
        auto dw = g_dual_ws.find(st);
        if (dw != g_dual_ws.end()) { if (dw->second.first) (void)hipFree(dw->second.first); g_dual_ws.erase(dw); }
        for (auto& dq : g_dq) {
            auto d = dq.find(st);
            if (d != dq.end()) { if (d->second.first) (void)hipFree(d->second.first); dq.erase(d); }
        }
    }
    release_fp8_workspace(st);
}
static int dequant_fp8(hipStream_t st, int which, const void* w8, const float* scale, int N, int K, const void** out) {
    const int KS = (K + 31) / 32, KSP = (KS + 1) / 2, NRG = (N + 15) / 16;
    const size_t bytes = (size_t)NRG * KS * 1024;
    void* dst;
    {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        auto& e = g_dq[which][st];
        if (e.second < bytes) {
            if (e.first) { SM_HIP(hipStreamSynchronize(st)); (void)hipFree(e.first); e.first = nullptr; e.second = 0; }
            SM_HIP(hipMalloc(&e.first, bytes));
            e.second = bytes;
        }
        dst = e.first;
    }
    const size_t chunks = (size_t)NRG * KSP * 64;
    dequant_fp8_kernel<<<(unsigned)((chunks + 255) / 256), 256, 0, st>>>((const u32x4*)w8, scale, N, KS, KSP, (u32x4*)dst, chunks);
    SM_LAUNCH_CHECK();
    *out = dst;
    return SM_OK;
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int WAVES>
static int launch_skinny_fp8(const LinArgs& a, bool xf32, bool split, bool dual, hipStream_t st) {
    dim3 grid(a.NRG), block(WAVES * 64);
    size_t sh = WAVES > 1 ? (size_t)WAVES * (dual ? 8 : 4) * 64 * sizeof(float) : 0;
#define SK(XF, SP, DU) skinny_fp8_kernel<WAVES, XF, SP, DU><<<grid, block, sh, st>>>(a, NoRope{})
    if (xf32) {
        if (split) { if (dual) SK(true, true, true); else SK(true, true, false); }
        else       { if (dual) SK(true, false, true); else SK(true, false, false); }
    } else {
        if (dual) SK(false, false, true); else SK(false, false, false);
    }
#undef SK
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// ---- 17..32 activation rows: weight-streaming with the activations SHARED through LDS.
// skinny_kernel gives every block one 16-row group and lets its waves split K, so each block re-reads (and re-converts) the
// whole activation matrix: at M = 32 fp32 that is 4 B of x per B of weight through L2/L1 and the kernel runs at 0.4-3 TB/s
// (time scaled with the activation bytes, tools/skinny_bench.py).  Here a block is 8 waves x 16 rows = 128 weight rows over ONE
// K slice: the x chunk (8 k-steps = 256 k, both 16-row column blocks) is converted once per block into B-fragment order in
// LDS (hi and, in precise mode, lo) and every wave multiplies it with its own weight rows; the K slices of the grid's second
// dimension are summed by splitk_reduce_kernel (fixed order), which also applies the epilogue.
// W8: weight-only fp8 (the packed image of pack_fp8_kernel: 16 bytes per lane = two k-steps), expanded to bf16 in registers -- half the weight bytes of
// the stream; the row scales stay in the epilogue of the slab pass (store4 / the norm pass), as for <= 16 rows.
// MB: 16-row blocks of activations a block holds (2: up to 32 rows; 4: up to 64 rows -- a batched decode step of 33..64 streams; 16-bit x only: the hi / lo
// pair of 64 fp32 rows would not fit the two LDS buffers).
template <bool XF32, bool SPLIT, bool DUAL, bool F16 = false, bool W8 = false, int MB = 2>
__global__ __launch_bounds__(512) void skinny_lds_kernel(LinArgs a, float* __restrict__ ws, float* __restrict__ ws2, int ks_per_split) {
    constexpr int KC = 8;                                   // k-steps per staged chunk
    static_assert(MB == 2 || (MB == 4 && !SPLIT), "skinny_lds_kernel: 64 rows only without the hi/lo split");
    __shared__ __attribute__((aligned(16))) bf16x8 xs[2][SPLIT ? 2 : 1][KC * MB * 64];      // [buf][hi/lo][(ks, mb, lane)]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int rg = blockIdx.x * 8 + wave;
    const bool wave_on = rg < a.NRG;
    const int KS = a.KS;
    const int ks0 = blockIdx.y * ks_per_split, ks1 = min(ks0 + ks_per_split, KS);
    const bf16x8* wp = a.w + (size_t)(wave_on ? rg : 0) * KS * 64 + lane;
    const bf16x8* wp2 = DUAL ? a.w2 + (size_t)(wave_on ? rg : 0) * KS * 64 + lane : nullptr;
    const int KSP = (KS + 1) >> 1;                                 // fp8 image: pairs of k-steps
    const u32x4* wq = (const u32x4*)a.w + (size_t)(wave_on ? rg : 0) * KSP * 64 + lane;
    const u32x4* wq2 = DUAL ? (const u32x4*)a.w2 + (size_t)(wave_on ? rg : 0) * KSP * 64 + lane : nullptr;
    f32x4 acc[MB], acc2[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) { acc[mb] = f32x4{0, 0, 0, 0}; acc2[mb] = f32x4{0, 0, 0, 0}; }

    // fill item = (ks_local, mb, lane'): 8 consecutive k of row mb*16 + (lane' & 15) -> one 16-byte B fragment (x2 in split mode).
    // Two halves: fill_load issues the (L2-resident) x loads, fill_store converts and writes LDS.  The weight loads of the
    // same chunk are issued BETWEEN the two: VMEM returns in order, so x loads queued behind the HBM weight loads would make
    // the conversion wait for the weights (measured: 7 us per chunk instead of ~3).
    constexpr int NIT = (KC * MB * 64) / 512;
    f32x4 xa[NIT], xb[NIT];
    bool xok[NIT];
    auto fill_load = [&](int kbase) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * 512;
            const int l2 = item & 63, mb = (item >> 6) & (MB - 1), ksl = item >> (MB == 4 ? 8 : 7);
            const int m = mb * 16 + (l2 & 15), ks = kbase + ksl;
            xok[it] = m < a.M && ks < ks1;
            xa[it] = f32x4{0, 0, 0, 0}; xb[it] = f32x4{0, 0, 0, 0};
            if (xok[it]) {
                const char* px = (const char*)a.x + ((size_t)m * a.ldx + (size_t)xcol(a, ks * 32 + (l2 >> 4) * 8)) * (XF32 ? 4 : 2);
                xa[it] = *(const f32x4*)px;                              // bf16 x: the 8 values are the 16 bytes of xa
                if (XF32) xb[it] = *(const f32x4*)(px + 16);
            }
        }
    };
    auto fill_store = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * 512;
            bf16x8 hi, lo;
            if (XF32) split_x<SPLIT, F16>(xa[it], xb[it], hi, lo);
            else hi = __builtin_bit_cast(bf16x8, xa[it]);
            xs[buf][0][item] = hi;
            if (SPLIT) xs[buf][SPLIT ? 1 : 0][item] = lo;
        }
    };
    // weights of chunk c+1 are requested before chunk c is multiplied (two register sets), the x chunk c+1 is converted into
    // the other LDS buffer meanwhile: one barrier per chunk, HBM latency hidden behind the previous chunk
    bf16x8 wa[2][W8 ? 1 : KC], wb[2][W8 ? 1 : KC];
    u32x4 qa[2][W8 ? KC / 2 : 1], qb[2][W8 ? KC / 2 : 1];          // fp8: one 16-byte load per lane = k-steps 2j and 2j + 1 (slices start at even k-steps)
    auto load_w = [&](int kb, int set) {
        if constexpr (W8) {
#pragma unroll
            for (int u = 0; u < KC / 2; ++u) {
                const int kp = min(kb + 2 * u, ks1 - 1) >> 1;
                qa[set][u] = __builtin_nontemporal_load(wq + (size_t)kp * 64);
                if (DUAL) qb[set][u] = __builtin_nontemporal_load(wq2 + (size_t)kp * 64);
            }
        } else {
#pragma unroll
            for (int u = 0; u < KC; ++u) {
                const int ks = min(kb + u, ks1 - 1);
                wa[set][u] = __builtin_nontemporal_load(wp + (size_t)ks * 64);
                if (DUAL) wb[set][u] = __builtin_nontemporal_load(wp2 + (size_t)ks * 64);
            }
        }
    };
    auto compute = [&](int kb, int buf, int set) {
#pragma unroll
        for (int u = 0; u < KC; ++u) {
            if (kb + u >= ks1) break;
            bf16x8 w0, w1;
            if constexpr (W8) {
                w0 = fp8x8_to_bf16(qa[set][u >> 1][(u & 1) * 2], qa[set][u >> 1][(u & 1) * 2 + 1]);
                if (DUAL) w1 = fp8x8_to_bf16(qb[set][u >> 1][(u & 1) * 2], qb[set][u >> 1][(u & 1) * 2 + 1]);
            } else {
                w0 = wa[set][u];
                if (DUAL) w1 = wb[set][u];
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const bf16x8 xh = xs[buf][0][(u * MB + mb) * 64 + lane];
                acc[mb] = mfma16<F16>(w0, xh, acc[mb]);
                if (DUAL) acc2[mb] = mfma16<F16>(w1, xh, acc2[mb]);
                if (SPLIT) {
                    const bf16x8 xl = xs[buf][SPLIT ? 1 : 0][(u * MB + mb) * 64 + lane];
                    acc[mb] = mfma16<F16>(w0, xl, acc[mb]);
                    if (DUAL) acc2[mb] = mfma16<F16>(w1, xl, acc2[mb]);
                }
            }
        }
    };
    fill_load(ks0);
    load_w(ks0, 0);
    fill_store(0);
    for (int kb = ks0; kb < ks1; kb += 2 * KC) {             // unrolled by two so that the register set is a literal
        __syncthreads();
        if (kb + KC < ks1) { fill_load(kb + KC); load_w(kb + KC, 1); fill_store(1); }
        compute(kb, 0, 0);
        if (kb + KC >= ks1) break;
        __syncthreads();
        if (kb + 2 * KC < ks1) { fill_load(kb + 2 * KC); load_w(kb + 2 * KC, 0); fill_store(0); }
        compute(kb + KC, 1, 1);
    }
    if (!wave_on) return;
    // raw partial sums of this K slice: slab [blockIdx.y][m][n]; lane (g, i) owns row m = mb*16 + i, columns rg*16 + g*4 .. +3
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = mb * 16 + i, n0 = rg * 16 + g * 4;
        if (m >= a.M || n0 >= a.N) continue;
        const size_t o = ((size_t)blockIdx.y * a.M + m) * a.N + n0;
        if (n0 + 3 < a.N) {
            *(f32x4*)(ws + o) = acc[mb];
            if (DUAL) *(f32x4*)(ws2 + o) = acc2[mb];
        } else {
            for (int r = 0; r < 4; ++r)
                if (n0 + r < a.N) { ws[o + r] = acc[mb][r]; if (DUAL) ws2[o + r] = acc2[mb][r]; }
        }
    }
}

static int launch_skinny_lds(const LinArgs& a, bool xf32, bool split, bool dual, hipStream_t st, const PostLn* ln = nullptr, bool* ln_done = nullptr, bool w8 = false) {
    const bool rows64 = a.M > 32;                            // 33..64 rows of 16-bit activations (the caller checked)
    const int nb = (a.NRG + 7) / 8;
    static int target = -1;
    if (target < 0) { const char* e = getenv("SM_SKINNY_LDS_BLOCKS"); target = e ? atoi(e) : 256; }
    int S = target / nb;                                     // one block per CU, ONE round: 224 blocks beat 336-448 (tail round)
    if (S < 1) S = 1;
    int per = (a.KS + S - 1) / S;
    per = (per + 7) / 8 * 8;                                 // whole chunks of 8 k-steps per slice
    S = (a.KS + per - 1) / per;
    float* ws = nullptr;
    const size_t slab = (size_t)S * a.M * a.N;
    int rc = splitk_workspace(st, slab * sizeof(float) * (dual ? 2 : 1), &ws);
    if (rc) return rc;
    float* ws2 = dual ? ws + slab : nullptr;
    const dim3 grid(nb, S);
#define SL(XF, SP, DU) skinny_lds_kernel<XF, SP, DU><<<grid, 512, 0, st>>>(a, ws, ws2, per)
#define SLH(XF, SP, DU) skinny_lds_kernel<XF, SP, DU, true><<<grid, 512, 0, st>>>(a, ws, ws2, per)
#define SL8(XF, SP, DU) skinny_lds_kernel<XF, SP, DU, false, true><<<grid, 512, 0, st>>>(a, ws, ws2, per)
    if (rows64) {
        if (w8) { if (dual) skinny_lds_kernel<false, false, true, false, true, 4><<<grid, 512, 0, st>>>(a, ws, ws2, per); else skinny_lds_kernel<false, false, false, false, true, 4><<<grid, 512, 0, st>>>(a, ws, ws2, per); }
        else if (a.f16) { if (dual) skinny_lds_kernel<false, false, true, true, false, 4><<<grid, 512, 0, st>>>(a, ws, ws2, per); else skinny_lds_kernel<false, false, false, true, false, 4><<<grid, 512, 0, st>>>(a, ws, ws2, per); }
        else { if (dual) skinny_lds_kernel<false, false, true, false, false, 4><<<grid, 512, 0, st>>>(a, ws, ws2, per); else skinny_lds_kernel<false, false, false, false, false, 4><<<grid, 512, 0, st>>>(a, ws, ws2, per); }
    } else if (w8) {       // weight-only fp8 (bf16 operands)
        if (xf32) {
            if (split) { if (dual) SL8(true, true, true); else SL8(true, true, false); }
            else       { if (dual) SL8(true, false, true); else SL8(true, false, false); }
        } else { if (dual) SL8(false, false, true); else SL8(false, false, false); }
    } else if (a.f16) {    // fp16 operands (weights fp16; activations fp16, or an fp16 hi/lo pair in precise mode)
        if (xf32) {
            if (split) { if (dual) SLH(true, true, true); else SLH(true, true, false); }
            else       { if (dual) SLH(true, false, true); else SLH(true, false, false); }
        } else { if (dual) SLH(false, false, true); else SLH(false, false, false); }
    } else if (xf32) {
        if (split) { if (dual) SL(true, true, true); else SL(true, true, false); }
        else       { if (dual) SL(true, false, true); else SL(true, false, false); }
    } else {
        if (dual) SL(false, false, true); else SL(false, false, false);
    }
#undef SL
#undef SLH
#undef SL8
    SM_LAUNCH_CHECK();
    if (ln && !dual && a.remap_in == 0 && (a.N & 255) == 0 && a.N <= 4096 && (a.ldo & 3) == 0 && (!a.residual || (a.ldr & 3) == 0)) {
        splitk_reduce_norm_rows_kernel<<<a.M, a.N / 4, 0, st>>>(a, ws, S, *ln);
        SM_LAUNCH_CHECK();
        *ln_done = true;
        return SM_OK;
    }
    const size_t nthr = (size_t)a.M * ((a.N + 3) / 4);
    splitk_reduce_kernel<<<(unsigned)((nthr + 255) / 256), 256, 0, st>>>(a, ws, ws2, S, a.N);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

template <int WAVES, int MB = 1>
static int launch_skinny(const LinArgs& a, bool xf32, bool split, bool dual, hipStream_t st) {
    dim3 grid(a.NRG), block(WAVES * 64);
    size_t sh = WAVES > 1 ? (size_t)WAVES * (dual ? 8 : 4) * MB * 64 * sizeof(float) : 0;
#define SK(XF, SP, DU) skinny_kernel<WAVES, XF, SP, DU, MB><<<grid, block, sh, st>>>(a, NoRope{})
#define SKH(XF, SP, DU) skinny_kernel<WAVES, XF, SP, DU, MB, false, false, true><<<grid, block, sh, st>>>(a, NoRope{})
    if (a.f16) {
        if (xf32) {
            if (split) { if (dual) SKH(true, true, true); else SKH(true, true, false); }
            else       { if (dual) SKH(true, false, true); else SKH(true, false, false); }
        } else { if (dual) SKH(false, false, true); else SKH(false, false, false); }
    } else if (xf32) {
        if (split) { if (dual) SK(true, true, true); else SK(true, true, false); }
        else       { if (dual) SK(true, false, true); else SK(true, false, false); }
    } else {
        if (dual) SK(false, false, true); else SK(false, false, false);
    }
#undef SK
#undef SKH
    SM_LAUNCH_CHECK();
    return SM_OK;
}

template <int WAVES>
static int launch_skinny_norm(const LinArgs& a, bool dual, bool fp8, hipStream_t st) {
    dim3 grid(a.NRG), block(WAVES * 64);
    const size_t sh = ((size_t)WAVES * (dual ? 8 : 4) * 64 + WAVES * 16) * sizeof(float) + (size_t)a.M * a.K * 2;
    if (fp8) {
        if (dual) skinny_fp8_kernel<WAVES, true, false, true, true><<<grid, block, sh, st>>>(a, NoRope{});
        else skinny_fp8_kernel<WAVES, true, false, false, true><<<grid, block, sh, st>>>(a, NoRope{});
    } else if (a.f16) {
        if (dual) skinny_kernel<WAVES, true, false, true, 1, true, false, true><<<grid, block, sh, st>>>(a, NoRope{});
        else skinny_kernel<WAVES, true, false, false, 1, true, false, true><<<grid, block, sh, st>>>(a, NoRope{});
    } else if (dual) skinny_kernel<WAVES, true, false, true, 1, true><<<grid, block, sh, st>>>(a, NoRope{});
    else skinny_kernel<WAVES, true, false, false, 1, true><<<grid, block, sh, st>>>(a, NoRope{});
    SM_LAUNCH_CHECK();
    return SM_OK;
}

static void fill_args(const sm_linear_t* p, LinArgs& a) {
    a.n_band = 0;
    a.w = (const bf16x8*)p->w;
    a.w2 = (const bf16x8*)p->w2;
    a.N = p->N; a.K = p->K;
    a.KS = (p->K + 31) / 32;
    a.NRG = (p->N + 15) / 16;
    a.x = p->x; a.M = p->M; a.ldx = p->ldx;
    a.bias = p->bias; a.act = p->act;
    a.residual = p->residual; a.ldr = p->ldr;
    a.out_f32 = p->out_f32; a.out_bf16 = (bf16_t*)p->out_bf16;
    a.ldo = p->ldo; a.ldo_bf16 = p->ldo_bf16;
    a.remap_in = p->remap_in; a.remap_out = p->remap_out; a.remap_off = p->remap_off;
    a.vt = (bf16_t*)p->vt; a.vt_n0 = p->vt_n0; a.vt_S = p->vt_S; a.vt_dh = p->vt_dh; a.vt_ld = p->vt_ld;
    const bool w8 = p->w_dtype == SM_W_FP8 || p->w_dtype == SM_W_FP8_MFMA;
    a.wscale = w8 ? p->w_scale : nullptr; a.wscale2 = w8 ? p->w2_scale : nullptr;
    a.ngamma = p->norm_gamma; a.neps = p->norm_eps;
    a.f16 = p->op_dtype == SM_OP_F16;
    a.xr_sh = a.xr_dh_sh = 0;
    if (p->x_rep > 1) { while ((1 << a.xr_sh) < p->x_rep) ++a.xr_sh; while ((1 << a.xr_dh_sh) < p->x_rep_dh) ++a.xr_dh_sh; }
    // LayerNorm folding (sm_linear validates the call): the producer's 16-bit copy travels in the out_bf16 slot
    a.fold_og = nullptr; a.fold_ostats = p->fold_stats_out;
    if (p->fold_stats_out) { a.fold_og = p->post_ln_gamma; a.out_bf16 = (bf16_t*)p->post_ln_out; a.ldo_bf16 = p->post_ln_ldo; }
    a.fold_istats = p->fold_stats_in; a.fold_ig = p->fold_g; a.fold_ic = p->fold_c;
    a.fold_itiles = p->fold_stats_in ? p->K / 256 : 0;
    a.fold_invd = p->fold_stats_in ? 1.0f / (float)p->K : 0.f;
    a.fold_eps = p->fold_eps;
}

// the decode step's q/k/v product with RoPE + KV append in the epilogue (SmRopeEpi, host.h): 16-bit or fp8 weights, head_dim
// 128, M <= 16 rows of fp32 activations, RMSNorm fused in front when p->norm_gamma is set
int sm_linear_qkv_rope(const sm_linear_t* p, const SmRopeEpi& re, void* stream) {
    const bool w8 = p && (p->w_dtype == SM_W_FP8 || p->w_dtype == SM_W_FP8_MFMA);
    SM_REQUIRE(p && p->w && p->x && !p->w2 && !p->bias && !p->residual && p->act == SM_ACT_NONE && (p->w_dtype == SM_W_BF16 || w8) && !p->vt &&
               p->remap_in == 0, "sm_linear_qkv_rope: plain q/k/v weights only");
    SM_REQUIRE(!w8 || (p->w_scale && ((size_t)p->w_scale & 15) == 0 && p->op_dtype == SM_OP_BF16), "sm_linear_qkv_rope: fp8 weights need 16-byte aligned row scales and bf16 operands");
    SM_REQUIRE(p->M > 0 && p->M <= 16 && p->M <= SM_MAX_SEG && p->x_dtype == SM_X_F32 && !p->precise && (p->K & 31) == 0 && (p->ldx & 3) == 0 &&
               p->N == (re.H + 2 * re.KV) * 128, "sm_linear_qkv_rope: M <= 16 fp32 rows, K %% 32 == 0, N = (H + 2 KV) * 128 (M=%d N=%d K=%d)", p->M, p->N, p->K);
    SM_REQUIRE(!p->norm_gamma || (long)p->M * p->K <= 16384, "sm_linear_qkv_rope: fused RMSNorm needs M*K <= 16384");
    SM_REQUIRE(re.cos_tab && re.sin_tab && re.q, "sm_linear_qkv_rope: null tables / q");
    LinArgs a;
    fill_args(p, a);
    hipStream_t st = (hipStream_t)stream;
    SmProfScope prof(SM_PROF_SKINNY, st);
    constexpr int WAVES = 8;
    SM_REQUIRE(a.KS >= WAVES * 4, "sm_linear_qkv_rope: K too small");
    const dim3 grid((re.H + 2 * re.KV) * 4), block(WAVES * 64);
    if (w8) {
        if (p->norm_gamma) {
            const size_t sh = ((size_t)WAVES * 8 * 64 + WAVES * 16) * sizeof(float) + (size_t)a.M * a.K * 2;
            skinny_fp8_kernel<WAVES, true, false, true, true, true><<<grid, block, sh, st>>>(a, re);
        } else {
            skinny_fp8_kernel<WAVES, true, false, true, false, true><<<grid, block, (size_t)WAVES * 8 * 64 * sizeof(float), st>>>(a, re);
        }
        SM_LAUNCH_CHECK();
        return SM_OK;
    }
    if (p->norm_gamma) {
        const size_t sh = ((size_t)WAVES * 8 * 64 + WAVES * 16) * sizeof(float) + (size_t)a.M * a.K * 2;
        if (a.f16) skinny_kernel<WAVES, true, false, true, 1, true, true, true><<<grid, block, sh, st>>>(a, re);
        else skinny_kernel<WAVES, true, false, true, 1, true, true><<<grid, block, sh, st>>>(a, re);
    } else {
        const size_t sh = (size_t)WAVES * 8 * 64 * sizeof(float);
        if (a.f16) skinny_kernel<WAVES, true, false, true, 1, false, true, true><<<grid, block, sh, st>>>(a, re);
        else skinny_kernel<WAVES, true, false, true, 1, false, true><<<grid, block, sh, st>>>(a, re);
    }
    SM_LAUNCH_CHECK();
    return SM_OK;
}

extern "C" int sm_norm_ex(const float* x, int M, int D, int ldx, const float* gamma, const float* beta, float eps, int post_act,
                          float* out_f32, void* out_bf16, int ldo, int op_dtype, void* stream);                        // vecops.hip
// 1 (default): fp8 weights only -- on bf16 weights the tiled / weight-streaming MFMA kernels are 5-7 % faster at 33..64 rows (same-box A/B: 64 streams 4.94 vs
// 5.32 ms per decode step), on fp8 weights the alternative is a bf16 expansion per call; 2: bf16 weights too (A/B); 0: off
int sm_skinny_lds64_on() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SM_SKINNY_LDS64"); on = e ? atoi(e) : 1; }
    return on;
}
static int linear_impl(const sm_linear_t* p, void* stream, bool* ln_done);
static thread_local SmSlabOut* g_leave_slabs = nullptr;          // set by sm_linear_leave_slabs (host.h) around one sm_linear call
int sm_linear_leave_slabs(const sm_linear_t* p, SmSlabOut* out, void* stream) {
    SM_REQUIRE(p && out, "sm_linear_leave_slabs: null args");
    out->ws = nullptr; out->S = 0; out->stride = 0;
    g_leave_slabs = out;
    const int rc = sm_linear(p, stream);
    g_leave_slabs = nullptr;
    return rc;
}
int sm_swiglu_ex(const float* gu, int M, int F, void* out, int f16, void* stream);           // vecops.hip
// fp32 [M][2F] gate | up rows of an SM_ACT_SWIGLU_DUAL call that does not run on the 256 x 256 kernel (per HIP stream, grown on demand; not the
// split-K slabs: the product itself may use those)
static int dual_workspace(hipStream_t st, size_t bytes, float** out) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto& e = g_dual_ws[st];
    if (e.second < bytes) {
        if (e.first) { SM_HIP(hipStreamSynchronize(st)); (void)hipFree(e.first); e.first = nullptr; e.second = 0; }
        SM_HIP(hipMalloc((void**)&e.first, bytes));
        e.second = bytes;
    }
    *out = e.first;
    return SM_OK;
}
// the tile rule of the LDS-tiled GEMM (linear_impl): 0 = 128 x 128, 256 / 257 = 256 x 256 (257: never persistent), 128 = 256 x 128
static int gemm_tile_choice(const sm_linear_t* p, int* hint_out = nullptr) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("SM_GEMM_TILE"); force = e ? atoi(e) : 0; }
    if (hint_out) *hint_out = p->tile_hint ? p->tile_hint : force;
    const int t256 = cdiv(p->M, 256) * cdiv(p->N, 256);
    const bool ok = !p->vt || p->vt_n0 % 256 == 0;
    int bn = (ok && t256 >= 192) ? 256 : 0;
    const int hint = p->tile_hint ? p->tile_hint : force;
    if (hint == 128) bn = 0;
    if (hint == 256128 && ok) bn = 128;
    if (hint == 256 && ok) bn = 256;
    if (hint == 2561 && ok) bn = 257;
    return bn;
}
// 33..128 rows of 16-bit activations on bf16 / fp16 (or weight-only fp8, expanded) weights: the weight-streaming MFMA kernel (wstream.hip).  Returns the number
// of K slabs (>= 1) and the k-steps per slab, or 0 when the call does not qualify.  SM_WSTREAM=0: off (the 128 x 128 tiled kernel, A/B)
static int wstream_slabs(const sm_linear_t* p, int* ksl_out) {
    static int use_ws = -1;
    if (use_ws < 0) { const char* e = getenv("SM_WSTREAM"); use_ws = e ? atoi(e) : 1; }
    const bool dual = p->act == SM_ACT_SWIGLU_DUAL;
    const int KS = (p->K + 31) / 32, NRG = (p->N + 15) / 16;
    if (!use_ws || p->M <= 32 || p->M > 128 || p->x_dtype != SM_X_BF16 || p->vt || p->remap_in || p->w2 || p->norm_gamma || p->tile_hint || (p->K & 31) || (p->ldx & 7) ||
        (KS % 8) || (p->w_dtype == SM_W_FP8_MFMA && (p->K & 127) == 0) || (dual && ((p->N & 31) || (p->ldo_bf16 & 3))) || p->x_rep > 1)
        return 0;
    const int nb = dual ? cdiv(NRG >> 1, 4) : cdiv(NRG, 8);
    // Wide products only (>= 192 column blocks: gate | up as SwiGLU-dual, lm_head): one block per CU streams the whole K loop.  The kernel also runs
    // narrow products as K slabs (SM_WSTREAM=2 enables: <= 8 slabs of whole register rings) -- measured no faster than the 128 x 128 tiled kernel there
    // (128 rows, same box: q|k|v 22.6 vs 16.6 us with 8 slabs, 17.0 with 4; o_proj 13.4 vs 13.9; down_proj 30.4 vs 31.0): with MFMAs, fragment reads and
    // barriers REMOVED the kernel streams at the same 4.5-4.8 TB/s (tools/experiments/wstream_probe.hip), i.e. both kernels sit on what 224-256 CUs pull
    // through 1-KiB row-group streams, and short slabs pay the ring fill twice
    int S = 1;
    if (nb < 192) {
        if (use_ws < 2 || dual) return 0;
        for (int s = 8; s >= 2; --s)
            if (nb * s <= 512 && KS % (8 * s) == 0) { S = s; break; }
        if (S == 1) return 0;
    }
    *ksl_out = KS / S;
    return S;
}
// Whole rounds + remainder (round 6; VERDICT r5 weak #4: 8 frames per call cost more per frame than 7, 16 more than 14).  A 16-bit-output product on the
// 256 x 256 kernel whose tile count is a little over a whole number of 256-CU rounds -- fc1 at 8 frames: 19 x 16 = 304 tiles = 1.19 rounds -- runs as TWO
// rounds, the second one on 48 CUs.  Rows are independent, so the call is cut at the last row tile that still fits whole rounds (4096 rows = 256 tiles = one
// round) and the remaining rows (520) go through sm_linear again, where their ~160 tiles of 128 x 128 are ONE partial round of the small kernel (two blocks
// per CU): one round + ~0.4 instead of two.  Only when the leftover is at most a QUARTER of a round and the epilogue is row-local (no residual / LayerNorm
// behind it / fold / remap).  Measured, same box, ms per call of F frames with the rule off / on at <= 140 leftover tiles (profiles/r06_rowsplit_ab.txt):
// 8 frames 5.51 -> 5.35, 10: 6.31 -> 6.19, 16: 8.72 -> 8.66, 20: 10.0 -> 9.90, but 14 frames 7.09 -> 7.37 (q|k|v: 128 leftover tiles = a third of the
// product on the small kernel) -- hence the quarter.  The cliffs themselves (8 frames cost more per frame than 7) are mostly NOT this round: out-proj / fc2 of
// such a call are 296 tiles of 128 x 128, and they stay.  SM_GEMM_ROWSPLIT=0 switches it off (A/B); returns 0 (no split) or the rows of the first part.
static int rowsplit_rows(const sm_linear_t* p) {
    static int on = -1, max_left = 64, n_cu = 0;
    if (on < 0) {
        const char* e = getenv("SM_GEMM_ROWSPLIT"); on = e ? atoi(e) : 1;
        const char* m = getenv("SM_GEMM_ROWSPLIT_MAXLEFT"); if (m) max_left = atoi(m);
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    }
    if (!on || n_cu < 8 || p->M <= 512 || (p->N & 255) || p->x_dtype != SM_X_BF16 || p->w_dtype != SM_W_BF16 || !p->out_bf16 || p->out_f32 || p->residual || p->vt || p->remap_in ||
        p->w2 || p->post_ln_gamma || p->fold_stats_in || p->fold_stats_out || p->norm_gamma || p->tile_hint || p->act == SM_ACT_SWIGLU_DUAL || p->x_rep > 1)
        return 0;
    const int bn = gemm_tile_choice(p);
    if (bn != 256) return 0;
    const int tiles_n = p->N / 256, tiles_m = cdiv(p->M, 256), T = tiles_m * tiles_n;
    const int R = T / n_cu, L = T % n_cu;
    if (R < 1 || L == 0 || L > max_left) return 0;
    const int rows_main = (R * n_cu) / tiles_n;          // row tiles that fit into R whole rounds
    if (rows_main < 1 || rows_main >= tiles_m) return 0;
    return rows_main * 256;
}

extern "C" int sm_linear(const sm_linear_t* p, void* stream) {
    SM_REQUIRE(p, "sm_linear: null args");
    if (const int m1 = rowsplit_rows(p)) {
        sm_linear_t a = *p, b = *p;
        a.M = m1;
        b.M = p->M - m1;
        b.x = (const char*)p->x + (size_t)m1 * p->ldx * 2;
        b.out_bf16 = (char*)p->out_bf16 + (size_t)m1 * p->ldo_bf16 * 2;
        int rc = sm_linear(&a, stream);          // (whole rounds: the rule does not fire again)
        return rc ? rc : sm_linear(&b, stream);
    }
    if (p->act == SM_ACT_SWIGLU_DUAL) {
        const int F = p->N >> 1;
        SM_REQUIRE(p->M > 16 && p->x_dtype == SM_X_BF16 && p->out_bf16 && !p->out_f32 && !p->residual && !p->vt && !p->w2 && p->remap_in == 0 && !p->post_ln_gamma &&
                   !p->norm_gamma && (p->N & 1) == 0 && p->ldo_bf16 >= F,
                   "sm_linear: SM_ACT_SWIGLU_DUAL needs M > 16, 16-bit x, one [gate | up] weight image (N even) and a 16-bit output [M][ldo_bf16 >= N / 2] only");
        static int dual_fuse = -1;                    // SM_SWIGLU_FUSE=0: always the product + the SwiGLU pass (A/B)
        if (dual_fuse < 0) { const char* e = getenv("SM_SWIGLU_FUSE"); dual_fuse = e ? atoi(e) : 1; }
        const int bn = gemm_tile_choice(p);
        const bool w8 = p->w_dtype == SM_W_FP8 || p->w_dtype == SM_W_FP8_MFMA;
        const bool fp8_mfma = p->w_dtype == SM_W_FP8_MFMA && p->M > 16 && (p->K & 127) == 0 && (p->ldx & 7) == 0;
        int ksl_ = 0;
        const bool fused = dual_fuse && !fp8_mfma && (!w8 || p->w_scale) &&
                           (((bn == 256 || bn == 257) && (p->N & 255) == 0 && (p->ldo_bf16 & 7) == 0 && ((uintptr_t)p->out_bf16 & 15) == 0) || wstream_slabs(p, &ksl_) == 1);
        // (N %% 256 == 0 -- whole tiles of 128 gate + 128 up rows -- is the 256 x 256 kernel's requirement only; the weight-streaming kernel takes N %% 32 == 0
        //  and the product + SwiGLU pass below any even N: a model with llm_mlp %% 128 == 64 prefills through it)
        if (!fused) {
            SM_REQUIRE(p->ldo_bf16 == F, "sm_linear: SM_ACT_SWIGLU_DUAL outside the 256 x 256 kernel writes a dense [M][N / 2] output (ldo_bf16=%d)", p->ldo_bf16);
            float* ws = nullptr;
            int rc = dual_workspace((hipStream_t)stream, (size_t)p->M * p->N * sizeof(float), &ws);
            if (rc) return rc;
            sm_linear_t q = *p;
            q.act = SM_ACT_NONE; q.out_bf16 = nullptr; q.out_f32 = ws; q.ldo = p->N;
            bool dummy = false;
            if ((rc = linear_impl(&q, stream, &dummy))) return rc;
            return sm_swiglu_ex(ws, p->M, F, p->out_bf16, p->op_dtype == SM_OP_F16 ? 1 : 0, stream);
        }
    }
    if (p->post_ln_gamma)
        SM_REQUIRE((p->post_ln_out || p->post_ln_out_f32) && p->out_f32 && !p->out_bf16 && p->remap_in == 0 && !p->vt && !p->w2 && p->post_ln_ldo >= p->N && (p->post_ln_ldo & 3) == 0,
                   "sm_linear: post-norm needs gamma, an output [M][post_ln_ldo >= N, %% 4 == 0] (16-bit and / or fp32), an fp32 product output (out_bf16 is not written by the fused "
                   "slab-sum + norm passes: leave it NULL) and plain rows");
    if (p->fold_stats_out || p->fold_stats_in) {
        // LayerNorm folding (sm_linear_t.fold_*): both sides live in the 256 x 256 tile kernels' epilogues
        const int bn = gemm_tile_choice(p);
        const bool w8f = p->w_dtype == SM_W_FP8 || p->w_dtype == SM_W_FP8_MFMA;
        SM_REQUIRE(!(p->fold_stats_out && p->fold_stats_in), "sm_linear: a call is the producer OR the consumer of a folded LayerNorm");
        SM_REQUIRE((bn == 256 || bn == 257) && p->M > 128 && (p->N & 255) == 0 && !w8f && p->x_dtype == SM_X_BF16 && !p->vt && p->remap_in == 0 && !p->w2 && !p->norm_gamma &&
                   p->x_rep <= 1, "sm_linear: LayerNorm folding needs the 256 x 256 tile kernel (>= 192 tiles or tile_hint SM_TILE_256, N %% 256 == 0, plain 16-bit operands; M=%d N=%d)", p->M, p->N);
        if (p->fold_stats_out)
            SM_REQUIRE(p->post_ln_gamma && p->post_ln_out && !p->post_ln_out_f32 && p->post_ln_act == SM_ACT_NONE && p->out_f32 && !p->out_bf16 && p->act == SM_ACT_NONE &&
                       (p->ldo & 3) == 0 && (p->post_ln_ldo & 7) == 0 && (!p->residual || (p->ldr & 3) == 0) &&
                       (((uintptr_t)p->out_f32 | (uintptr_t)p->post_ln_out | (uintptr_t)p->residual | (uintptr_t)p->post_ln_gamma | (uintptr_t)p->bias) & 15) == 0 && ((uintptr_t)p->fold_stats_out & 7) == 0,
                       "sm_linear: the producer of a folded LayerNorm is an fp32 (+ residual) product with post_ln_gamma / post_ln_out (16-byte aligned rows)");
        else
            SM_REQUIRE(p->fold_g && p->fold_c && (p->K & 255) == 0 && p->out_bf16 && !p->out_f32 && !p->residual && !p->post_ln_gamma && (p->ldo_bf16 & 7) == 0 && ((uintptr_t)p->out_bf16 & 15) == 0 &&
                       (((uintptr_t)p->fold_g | (uintptr_t)p->fold_c | (uintptr_t)p->fold_stats_in) & 15) == 0 && (p->act == SM_ACT_NONE || p->act == SM_ACT_QUICK_GELU),
                       "sm_linear: the consumer of a folded LayerNorm is a 16-bit-output product (K %% 256 == 0, act none / quick_gelu) with fold_g / fold_c");
    }
    if (p->x_rep > 1)
        SM_REQUIRE(p->M <= 32 && p->x_dtype == SM_X_F32 && p->w_dtype == SM_W_BF16 && !p->norm_gamma && (p->x_rep & (p->x_rep - 1)) == 0 && p->x_rep_dh >= 8 &&
                   (p->x_rep_dh & (p->x_rep_dh - 1)) == 0 && p->K % (p->x_rep * p->x_rep_dh) == 0 && p->ldx >= p->K / p->x_rep,
                   "sm_linear: x_rep needs the weight-streaming path (M <= 32, fp32 x, 16-bit weights), powers of two, K a multiple of x_rep * x_rep_dh");
    bool ln_done = false;
    int rc = linear_impl(p, stream, &ln_done);
    if (rc || !p->post_ln_gamma || ln_done) return rc;
    return sm_norm_ex(p->out_f32, p->M, p->N, p->ldo, p->post_ln_gamma, p->post_ln_beta, p->post_ln_eps, p->post_ln_act, p->post_ln_out_f32, p->post_ln_out,
                      p->post_ln_ldo, p->op_dtype, stream);
}
static int linear_impl(const sm_linear_t* p, void* stream, bool* ln_done) {
    SM_REQUIRE(p && p->w && p->x, "sm_linear: null w/x");
    SM_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, "sm_linear: bad dims M=%d N=%d K=%d", p->M, p->N, p->K);
    SM_REQUIRE(p->out_f32 || p->out_bf16 || p->vt, "sm_linear: no output");
    LinArgs a;
    fill_args(p, a);
    const bool w8 = p->w_dtype == SM_W_FP8 || p->w_dtype == SM_W_FP8_MFMA;
    SM_REQUIRE(p->op_dtype == SM_OP_BF16 || p->op_dtype == SM_OP_F16, "sm_linear: op_dtype must be SM_OP_BF16 or SM_OP_F16");
    SM_REQUIRE(!a.f16 || !w8, "sm_linear: fp16 operands exclude fp8 weights");
    SM_REQUIRE(!a.f16 || p->M <= 32 || (p->x_dtype == SM_X_BF16 && (!p->w2 || p->M <= 64) && !p->norm_gamma),
               "sm_linear: above 32 rows fp16 operands run on the tiled GEMM (16-bit x, single weight image, no fused norm)");
    SM_REQUIRE(!p->norm_gamma || (p->M <= 16 && (long)p->M * p->K <= 16384 && p->x_dtype == SM_X_F32 && !p->precise && (p->K & 31) == 0 && (p->ldx & 3) == 0),
               "sm_linear: fused RMSNorm needs M <= 16, M*K <= 16384 (the normalised rows live in LDS), fp32 x (precise = 0), K %% 32 == 0 (M=%d K=%d)", p->M, p->K);
    SM_REQUIRE(!w8 || (p->w_scale && (!p->w2 || p->w2_scale)), "sm_linear: fp8 weights need their row scales");
    // 33..64 rows of 16-bit activations on fp8 weights: the LDS-shared weight-streaming kernel with four 16-row blocks (a batched decode step of 33..64 streams)
    const bool w8_any = p->w_dtype == SM_W_FP8 || p->w_dtype == SM_W_FP8_MFMA;
    const bool lds64 = (sm_skinny_lds64_on() >= 2 || (sm_skinny_lds64_on() == 1 && w8_any)) && p->M > 32 && p->M <= 64 && p->x_dtype == SM_X_BF16 && !p->precise && !p->norm_gamma && ((p->N & 3) == 0 || p->N < 4) && a.KS >= 8 &&
                       p->remap_in == 0 && !p->vt && (p->ldx % 8) == 0 && p->act != SM_ACT_SWIGLU_DUAL;
    SM_REQUIRE(!p->w2 || p->M <= 32 || lds64, "sm_linear: dual weights only on the weight-streaming path (M <= 32; 16-bit rows up to 64)");
    SM_REQUIRE(p->x_rep > 1 || p->ldx >= a.KS * 32, "sm_linear: ldx=%d must cover K padded to 32 (%d)", p->ldx, a.KS * 32);
    SM_REQUIRE(!p->vt || (p->vt_dh > 0 && p->vt_S > 0 && (p->N - p->vt_n0) % p->vt_dh == 0), "sm_linear: bad vt args");
    hipStream_t st = (hipStream_t)stream;
    const bool xf32 = p->x_dtype == SM_X_F32;
    if (p->w_dtype == SM_W_FP8_MFMA && p->M > 16 && (p->K & 127) == 0 && (p->ldx & 7) == 0 && !p->precise && !p->w2 && !xf32 && !p->vt && p->remap_in == 0) {
        // fp8 x fp8 on the matrix pipe: activation rows quantised to e4m3, no bf16 expansion of the weights (gemm_fp8.hip)
        SmProfScope prof(SM_PROF_GEMM, st, ((long long)p->N << 32) | (unsigned)p->K);
        return launch_gemm_fp8(a, st);
    }
    // 17..32 rows on fp8 weights: the LDS-shared weight-streaming kernel reads the fp8 image itself (same conditions as its bf16 dispatch below)
    static int w8_lds_on = -1;
    if (w8_lds_on < 0) { const char* e = getenv("SM_FP8_LDS"); w8_lds_on = e ? atoi(e) : 1; }
    // (N < 4 -- the gate head's two rows -- too: every slab store of such a product takes the element path, and the bf16 expansion it would fall back to
    //  rounds q * s to bf16: 4e-3 on the gate logits where the streamed image gives 3e-5)
    static int lds_min_m8 = -1;                   // (the same threshold as the bf16 dispatch below: SM_SKINNY_LDS_MINM)
    if (lds_min_m8 < 0) { const char* e = getenv("SM_SKINNY_LDS_MINM"); lds_min_m8 = e ? atoi(e) : 10; }
    // (10..16 rows on fp8 weights: only with fp32 activations -- the connector / gate pass 568 -> 460 us at 16 rows; with 16-bit activations, a 16-stream decode step,
    //  the <= 16-row fp8 kernels are faster: 2.97 vs 3.17 ms)
    const bool w8_lds = w8 && w8_lds_on && (p->M > 16 || (p->M >= lds_min_m8 && p->x_dtype == SM_X_F32)) && p->M <= 32 && !p->norm_gamma && ((p->N & 3) == 0 || p->N < 4) && a.KS >= 8 && p->remap_in == 0 && !p->vt;
    if (w8 && p->M > 16 && !w8_lds && !lds64) {
        // the fp8 kernels are weight-streaming only (one MFMA column block): more rows expand the weights to a bf16 scratch
        // image (row scale folded in) and take the bf16 kernels -- 1.5x the weight bytes once per call instead of M/16 passes
        const void *d0 = nullptr, *d1 = nullptr;
        int rc = dequant_fp8(st, 0, p->w, p->w_scale, p->N, p->K, &d0);
        if (!rc && p->w2) rc = dequant_fp8(st, 1, p->w2, p->w2_scale, p->N, p->K, &d1);
        if (rc) return rc;
        a.w = (const bf16x8*)d0;
        if (p->w2) a.w2 = (const bf16x8*)d1;
        a.wscale = a.wscale2 = nullptr;
    }
    const bool w8k = w8 && (p->M <= 16 || w8_lds || lds64);   // fp8 kernels in use
    if (p->M <= 32) {
        SM_REQUIRE(!xf32 || (p->ldx % 4 == 0), "sm_linear: fp32 x needs ldx %% 4 == 0");
        SM_REQUIRE(xf32 || (p->ldx % 8 == 0), "sm_linear: bf16 x needs ldx %% 8 == 0");
        const bool split = xf32 && p->precise;
        const bool dual = p->w2 != nullptr;
        SmProfScope prof(SM_PROF_SKINNY, st);
        if (p->norm_gamma) {
            // the dual (gate / up) kernel as 4-wave blocks: 896 blocks of 8 waves are 1.75 rounds of the 512 block slots of the chip,
            // smaller blocks leave a shorter tail (Mistral-7B decode 343.5 -> 346.8 tokens/s, fp8 519 -> 524); the single-matrix
            // kernels (lm_head) measured unchanged.  SM_NORM_WAVES=8 restores the 8-wave blocks (A/B)
            static int nw = -1;
            if (nw < 0) { const char* e = getenv("SM_NORM_WAVES"); nw = e ? atoi(e) : 0; }
            if (dual && nw != 8 && a.KS >= 8) return launch_skinny_norm<4>(a, dual, w8k, st);
            if (a.KS >= 32) return launch_skinny_norm<8>(a, dual, w8k, st);
            if (a.KS >= 8) return launch_skinny_norm<4>(a, dual, w8k, st);
            return launch_skinny_norm<1>(a, dual, w8k, st);
        }
        if (w8_lds) {
            static int ln_fuse8 = -1;
            if (ln_fuse8 < 0) { const char* e = getenv("SM_POST_LN_FUSE"); ln_fuse8 = e ? atoi(e) : 1; }
            const PostLn ln = {p->post_ln_gamma, p->post_ln_beta, p->post_ln_eps, (bf16_t*)p->post_ln_out, p->post_ln_ldo, p->post_ln_out_f32, p->post_ln_act};
            return launch_skinny_lds(a, xf32, split, dual, st, (ln_fuse8 && p->post_ln_gamma && p->out_f32) ? &ln : nullptr, ln_done, true);
        }
        if (w8k) {
            static int f8w = -1;                      // SM_FP8_WAVES=4|8|16: tuning override for the fp8 weight-streaming kernels
            if (f8w < 0) { const char* e = getenv("SM_FP8_WAVES"); f8w = e ? atoi(e) : 0; }
            if (f8w == 4 && a.KS >= 8) return launch_skinny_fp8<4>(a, xf32, split, dual, st);
            if (f8w == 8 && a.KS >= 32) return launch_skinny_fp8<8>(a, xf32, split, dual, st);
            if (f8w == 16 && a.KS >= 64 && !dual) return launch_skinny_fp8<16>(a, xf32, split, dual, st);
            if (a.KS >= 32) return launch_skinny_fp8<8>(a, xf32, split, dual, st);        // with the register ring 8 waves beat 16 (Mistral-7B decode 515 vs 508 tokens/s)
            if (a.KS >= 8) return launch_skinny_fp8<4>(a, xf32, split, dual, st);
            return launch_skinny_fp8<1>(a, xf32, split, dual, st);
        }
        // enough waves per row-group to keep >= ~32 KiB of weight loads in flight per CU
        {   static int fw = -1;                       // SM_SKINNY_WAVES=4|8|16: tuning override (tools/decode_bench.py)
            if (fw < 0) { const char* e = getenv("SM_SKINNY_WAVES"); fw = e ? atoi(e) : 0; }
            if (fw == 4 && a.KS >= 4 && p->M <= 16) return launch_skinny<4>(a, xf32, split, dual, st);
            if (fw == 8 && a.KS >= 8 && p->M <= 16) return launch_skinny<8>(a, xf32, split, dual, st);
            if (fw == 16 && a.KS >= 16 && !dual && p->M <= 16) return launch_skinny<16>(a, xf32, split, dual, st);
        }
        // 17..32 rows, SMALL weights (the gate's V, the connector's x_proj / dt_proj: 1-8 MB): the K-slice kernel + its slab-sum launch are two
        // launches of ~5 us for ~1 us of traffic; the two-column-block kernel does it in one (activations re-read per block from L2).
        // MEASURED AND LEFT OFF (SM_SKINNY_DIRECT_MB=n enables it for weights up to n MB): the 28-row connector + gate pass 588 -> 636 us with
        // n = 12 or 40 -- one block per 16 output rows walking all of K is a longer latency chain than the launch it saves.
        {
            static int direct_mb = -1;
            if (direct_mb < 0) { const char* e = getenv("SM_SKINNY_DIRECT_MB"); direct_mb = e ? atoi(e) : 0; }
            if (direct_mb > 0 && p->M > 16 && !p->post_ln_gamma && !w8k && (size_t)p->N * p->K * 2 <= (size_t)direct_mb * 1048576 && p->remap_in == 0 && !p->vt) {
                if (a.KS >= 32) return launch_skinny<8, 2>(a, xf32, split, dual, st);
                if (a.KS >= 8) return launch_skinny<4, 2>(a, xf32, split, dual, st);
                return launch_skinny<1, 2>(a, xf32, split, dual, st);
            }
        }
        static int lds_min_m = -1;                    // SM_SKINNY_LDS_MINM: smallest M sent to the LDS-shared kernel (tuning)
        // 10 (round 5; was 17): same-box scan of the connector + gate pass -- 10 / 12 / 14 / 16 rows 521 / 557 / 586 / 625 us on the <= 16-row kernels against
        // 502 / 513 / 511 / 518 through this one (8 rows and fewer: no difference); a 16-stream decode step 4.08 -> 3.84 ms, 12 streams equal
        //   With 16-bit activations (a batched decode step) the crossover sits higher: 10 / 12 streams 3.77 / 3.79 ms through this kernel against ~3.62 / 3.77 on
        //   the <= 16-row kernels, 16 streams 3.85 against 4.08 -- from 13 rows there (SM_SKINNY_LDS_MINM sets both).
        static int lds_min_m16 = 13;
        if (lds_min_m < 0) { const char* e = getenv("SM_SKINNY_LDS_MINM"); lds_min_m = e ? atoi(e) : 10; if (e) lds_min_m16 = lds_min_m; }
        {
            static int use_lds = -1;
            if (use_lds < 0) { const char* e = getenv("SM_SKINNY_LDS"); use_lds = e ? atoi(e) : 1; }
            if (use_lds && p->M >= (xf32 ? lds_min_m : lds_min_m16) && (p->N & 3) == 0 && a.KS >= 8 && p->remap_in == 0 && !p->vt) {
                static int ln_fuse_rows = -1;
                if (ln_fuse_rows < 0) { const char* e = getenv("SM_POST_LN_FUSE"); ln_fuse_rows = e ? atoi(e) : 1; }
                const PostLn ln = {p->post_ln_gamma, p->post_ln_beta, p->post_ln_eps, (bf16_t*)p->post_ln_out, p->post_ln_ldo, p->post_ln_out_f32, p->post_ln_act};
                return launch_skinny_lds(a, xf32, split, dual, st, (ln_fuse_rows && p->post_ln_gamma && p->out_f32) ? &ln : nullptr, ln_done);
            }
        }
        if (p->M > 16) {       // fallback for 17..32 rows: two MFMA column blocks share every weight load, activations re-read per block
            if (a.KS >= 32) return launch_skinny<8, 2>(a, xf32, split, dual, st);
            if (a.KS >= 8) return launch_skinny<4, 2>(a, xf32, split, dual, st);
            return launch_skinny<1, 2>(a, xf32, split, dual, st);
        }
        // measured on the Mistral-7B decode step: 8 waves per row-group (2 blocks/CU) 297 tok/s, 16 waves 291, 4 waves 265
        if (a.KS >= 32) return launch_skinny<8>(a, xf32, split, dual, st);
        if (a.KS >= 8) return launch_skinny<4>(a, xf32, split, dual, st);
        return launch_skinny<1>(a, xf32, split, dual, st);
    }
    if (lds64) {
        SmProfScope prof(SM_PROF_SKINNY, st);
        static int ln_fuse64 = -1;
        if (ln_fuse64 < 0) { const char* e = getenv("SM_POST_LN_FUSE"); ln_fuse64 = e ? atoi(e) : 1; }
        const PostLn ln = {p->post_ln_gamma, p->post_ln_beta, p->post_ln_eps, (bf16_t*)p->post_ln_out, p->post_ln_ldo, p->post_ln_out_f32, p->post_ln_act};
        return launch_skinny_lds(a, false, false, p->w2 != nullptr, st, (ln_fuse64 && p->post_ln_gamma && p->out_f32) ? &ln : nullptr, ln_done, w8);
    }
    SM_REQUIRE(!xf32, "sm_linear: the tiled GEMM takes bf16 activations (M=%d > 32)", p->M);

    SM_REQUIRE(p->ldx % 8 == 0, "sm_linear: bf16 x needs ldx %% 8 == 0");
    SM_REQUIRE(!p->vt || (p->vt_n0 % GEMM_BN == 0 && !p->residual && p->remap_in == 0), "sm_linear: vt_n0 must be a multiple of %d on the GEMM path", GEMM_BN);
    {   // 33..128 rows: a weight stream (wstream.hip)
        int ksl = 0;
        const int S = wstream_slabs(p, &ksl);
        if (S >= 1) {
            SmProfScope prof(SM_PROF_GEMM, st, ((long long)p->N << 32) | (unsigned)p->K);
            if (S == 1) return launch_wstream(a, st, nullptr, 1, ksl);
            float* ws = nullptr;
            int rc = splitk_workspace(st, (size_t)S * p->M * p->N * sizeof(float), &ws);
            if (rc) return rc;
            if ((rc = launch_wstream(a, st, ws, S, ksl))) return rc;
            static int ln_fuse_ws = -1;
            if (ln_fuse_ws < 0) { const char* e = getenv("SM_POST_LN_FUSE"); ln_fuse_ws = e ? atoi(e) : 1; }
            if (p->post_ln_gamma && ln_fuse_ws && (p->N & 255) == 0 && p->N <= 4096 && (p->ldo & 3) == 0 && (!p->residual || (p->ldr & 3) == 0) && !a.wscale) {
                const PostLn ln = {p->post_ln_gamma, p->post_ln_beta, p->post_ln_eps, (bf16_t*)p->post_ln_out, p->post_ln_ldo, p->post_ln_out_f32, p->post_ln_act};
                splitk_reduce_norm_rows_kernel<<<p->M, p->N / 4, 0, st>>>(a, ws, S, ln);
                SM_LAUNCH_CHECK();
                *ln_done = true;
                return SM_OK;
            }
            return launch_splitk_reduce(a, ws, S, p->N, st);
        }
    }
    // tile choice: 256x256 (one 8-wave block per CU, 128 FLOP per L2 byte) once its grid fills >= 3/4 of the chip,
    // else 128x128 (two blocks per CU); SM_GEMM_TILE=128|256128|256 overrides (tools/gemm_bench.py).  Measured on the
    // ViT shapes: B=28 frames (M=16156) 739 vs 706 TFLOP/s, B=14 565 vs 654 -> the threshold.
    {
        int hint = 0;
        const int bn = gemm_tile_choice(p, &hint);
        SM_REQUIRE(p->act != SM_ACT_SWIGLU_DUAL || bn == 256 || bn == 257, "sm_linear: internal: SwiGLU-dual outside the 256 x 256 kernel");
        // 64..128 tiles of 256 x 256 with >= 768 rows (whole row tiles: a 128-row batched decode step with N = 28672 also has 112 tiles and stays on the
        // 128 x 128 kernel), a long K loop and wide rows (the o / down products of an LLM prefill chunk: M = 1024..2048, N = 4096): split-K
        // slabs on the 256 x 256 kernel so that every CU multiplies (2048 rows, same box: down 275 -> ~215 us, o 85 -> ~77 us against the 128 x 128
        // kernel's 512 tiles), then ONE pass sums the slabs in slab order, applies bias / activation / residual and -- when the call carries one
        // -- the RMSNorm / LayerNorm of the finished row (sm_linear_t.post_ln_*).  SM_GEMM256_SPLITK=0: off (A/B)
        static int sk256 = -1;
        if (sk256 < 0) { const char* e = getenv("SM_GEMM256_SPLITK"); sk256 = e ? atoi(e) : 1; }
        const int t256 = cdiv(p->M, 256) * cdiv(p->N, 256);
        static int sk_minrows = 768, sk_mint = 64, sk_smax = 4;                 // SM_GEMM256_SPLITK_MINROWS / _MINTILES / _SMAX: the rule's thresholds (A/B)
        static bool sk_env = false;
        if (!sk_env) { const char* e = getenv("SM_GEMM256_SPLITK_MINROWS"); if (e) sk_minrows = atoi(e); e = getenv("SM_GEMM256_SPLITK_MINTILES"); if (e) sk_mint = atoi(e);
                       e = getenv("SM_GEMM256_SPLITK_SMAX"); if (e) sk_smax = atoi(e); sk_env = true; }
        if (sk256 && bn == 0 && hint == 0 && p->M >= sk_minrows && t256 >= sk_mint && t256 <= 128 && p->N >= 2048 && (p->N & 255) == 0 && p->K >= 4096 && p->out_f32 && !p->out_bf16 && !p->vt &&
            p->remap_in == 0 && (p->ldo & 3) == 0 && (!p->residual || (p->ldr & 3) == 0) && !a.wscale) {
            int S = 256 / t256;
            if (S > sk_smax) S = sk_smax;
            while (S > 1 && a.KS % S) --S;
            if (S >= 2) {
                float* ws = nullptr;
                int rc = splitk_workspace(st, (size_t)S * p->M * p->N * sizeof(float), &ws);
                if (rc) return rc;
                LinArgs b = a;
                b.out_f32 = ws; b.ldo = p->N; b.out_bf16 = nullptr; b.bias = nullptr; b.residual = nullptr; b.act = SM_ACT_NONE;
                {   SmProfScope prof(SM_PROF_GEMM, st, ((long long)p->N << 32) | (unsigned)p->K);
                    if ((rc = launch_gemm256(b, SM_ACT_NONE, 256, st, S))) return rc; }
                if (p->post_ln_gamma && p->N <= 4096) {
                    const PostLn ln = {p->post_ln_gamma, p->post_ln_beta, p->post_ln_eps, (bf16_t*)p->post_ln_out, p->post_ln_ldo, p->post_ln_out_f32, p->post_ln_act};
                    splitk_reduce_norm_rows_kernel<<<p->M, p->N / 4, 0, st>>>(a, ws, S, ln);
                    SM_LAUNCH_CHECK();
                    *ln_done = true;
                    return SM_OK;
                }
                if (g_leave_slabs && !p->bias && !p->residual && p->act == SM_ACT_NONE && !p->post_ln_gamma) {
                    g_leave_slabs->ws = ws; g_leave_slabs->S = S; g_leave_slabs->stride = (size_t)p->M * p->N;        // the caller's next kernel sums them on load
                    return SM_OK;
                }
                return launch_splitk_reduce(a, ws, S, p->N, st);
            }
        }
        if (bn) {
            SmProfScope prof(SM_PROF_GEMM, st, ((long long)p->N << 32) | (unsigned)p->K);
            if (p->fold_stats_out) *ln_done = true;          // the folded LayerNorm's producer: no norm launch behind it (the consumer applies mu / rstd)
            return launch_gemm256(a, p->act, bn, st);
        }
        SM_REQUIRE(!p->fold_stats_out && !p->fold_stats_in, "sm_linear: internal: LayerNorm folding outside the 256 x 256 kernel");
    }
    SM_REQUIRE((a.KS & 1) == 0, "sm_linear: the 128x128 GEMM needs K padded to a multiple of 64 (K=%d)", p->K);
    int tiles_m = cdiv(p->M, GEMM_BM), tiles_n = cdiv(p->N, GEMM_BN);
    static bool attr_set = false;
    if (!attr_set) {
#define GEMM_ATTR(ACT)                                                                                                       \
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<ACT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GEMM_STAGE_BYTES)); \
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<ACT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GEMM_STAGE_BYTES))
        GEMM_ATTR(0); GEMM_ATTR(1); GEMM_ATTR(-1);
#undef GEMM_ATTR
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<0, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GEMM_STAGE_BYTES));
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<1, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GEMM_STAGE_BYTES));
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<-1, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GEMM_STAGE_BYTES));
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<0, 8, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * GEMM_STAGE_BYTES));
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<1, 8, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * GEMM_STAGE_BYTES));
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<-1, 8, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * GEMM_STAGE_BYTES));
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<0, 8, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * GEMM_STAGE_BYTES));
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<1, 8, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * GEMM_STAGE_BYTES));
        SM_HIP(hipFuncSetAttribute((const void*)gemm_kernel<-1, 8, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * GEMM_STAGE_BYTES));
        attr_set = true;
    }
    // Few tiles (a single frame through the ViT: 40-160 tiles; LLM prefill chunks): split-K so that ~256 blocks exist, but
    // only for long K loops -- measured per shape (tools/gemm_small_sweep.py): at K = 1024 the reduce pass costs more than the
    // shorter loop saves (ViT qkv 14.5 vs 21.2 us), at K = 4096 / 40 tiles 4 slabs win (38.4 -> 20.5 us).  SM_SPLITK=n forces n.
    static int force_s = -1, use_w8 = -1;
    if (force_s < 0) { const char* e = getenv("SM_SPLITK"); force_s = e ? atoi(e) : 0; }
    if (use_w8 < 0) { const char* e = getenv("SM_GEMM_W8"); use_w8 = e ? atoi(e) : 2; }   // 0: 4-wave blocks, 1: 8 waves when blocks <= 256, 2: always
    const int tiles = tiles_m * tiles_n, KTall = a.KS >> 1;
    {   // tile order of the 128 x 128 kernel (SM_GEMM128_NBAND = 0 | 1 forces, default: by shape)
        static int nband = -2;
        if (nband == -2) { const char* e = getenv("SM_GEMM128_NBAND"); nband = e ? atoi(e) : -1; }
        a.n_band = nband >= 0 ? nband : (tiles_m > 1 && tiles_m < tiles_n ? 1 : 0);
    }
    int S = tiles <= 128 ? 256 / tiles : 1;
    // up to 128 rows (a batched decode step of 33..128 streams: one row tile, the product is a weight stream) more slabs pay: every CU should
    // pull weights, and a slab of 128 rows is 2 MB (SM_SPLITK_SMALLM_MAX, default 8; 4 = the rule of the larger shapes)
    static int smallm_max = -1;
    if (smallm_max < 0) { const char* e = getenv("SM_SPLITK_SMALLM_MAX"); smallm_max = e ? atoi(e) : 8; }
    const int s_cap = p->M <= 128 ? smallm_max : 4;
    if (S > s_cap) S = s_cap;
    while (S > 1 && KTall / S < (p->M <= 128 ? 8 : 16)) --S;
    // a post-LN call pays for a second pass anyway (the LayerNorm of the finished rows): as slabs + the fused slab-sum / LayerNorm pass
    // the product costs one launch less than GEMM + LayerNorm, so slabs pay even for short K loops (out-proj, K = 1024: 4 k-tiles each)
    static int ln_fuse = -1, ln_smax = 6;
    static int ln_maxtiles = 128;
    if (ln_fuse < 0) { const char* e = getenv("SM_POST_LN_FUSE"); ln_fuse = e ? atoi(e) : 1; const char* m = getenv("SM_POST_LN_SMAX"); if (m) ln_smax = atoi(m);
                       const char* t = getenv("SM_POST_LN_MAXTILES"); if (t) ln_maxtiles = atoi(t); }
    const bool ln_ok = ln_fuse && p->post_ln_gamma && p->post_ln_beta && p->post_ln_out && !p->post_ln_out_f32 && p->post_ln_act == SM_ACT_NONE && p->N == 1024 && tiles <= ln_maxtiles && p->act == SM_ACT_NONE && !w8 && (p->ldo & 3) == 0 &&
                       (!p->residual || (p->ldr & 3) == 0) && p->M <= 4096 && ln_smax >= 2;
    if (ln_ok) {
        S = tiles <= 128 ? 256 / tiles : 2;
        if (S > ln_smax) S = ln_smax;
        if (S > 8) S = 8;
        while (S > 2 && KTall / S < 4) --S;
        if (S < 2) S = 2;
    }
    if (force_s > 0 && !ln_ok) S = force_s;
    if (S > KTall) S = KTall;
    if (S < 1 || p->vt || (p->N & 3) || p->M > 4096) S = 1;
    const bool wv8 = use_w8 == 2 || (use_w8 == 1 && tiles * S <= 256);
    static int use_ring = -1;                       // SM_GEMM_RING=0: the two-stage loop everywhere (A/B)
    if (use_ring < 0) { const char* e = getenv("SM_GEMM_RING"); use_ring = e ? atoi(e) : 1; }
    // the ring kernel owns its CU (128 KiB of LDS): it wins where the whole grid is ONE partial round (one frame: QKV 14.9 -> 13.8 us
    // in situ, its k-loop 0.83 -> 0.41 us per k-tile L2-warm), and loses where the two-stage kernel's second resident block per CU hides
    // prologues and epilogues across rounds (4 / 8 frames per call: 3.88 -> 3.98 / 5.48 -> 5.70 ms)
    const bool ring = use_ring && (wv8 || a.f16) && (tiles * S <= 256 || use_ring == 2);
    SmProfScope prof(SM_PROF_GEMM, st, ((long long)p->N << 32) | (unsigned)p->K);
#define GEMM_LAUNCH(ACT, ARGS, GRID)                                                                                         \
    do {                                                                                                                     \
        if (ring && a.f16) gemm_kernel<ACT, 8, true, 4><<<GRID, 512, 4 * GEMM_STAGE_BYTES, st>>>(ARGS, tiles_m, tiles_n);    \
        else if (ring) gemm_kernel<ACT, 8, false, 4><<<GRID, 512, 4 * GEMM_STAGE_BYTES, st>>>(ARGS, tiles_m, tiles_n);       \
        else if (a.f16) gemm_kernel<ACT, 8, true><<<GRID, 512, 2 * GEMM_STAGE_BYTES, st>>>(ARGS, tiles_m, tiles_n);          \
        else if (wv8) gemm_kernel<ACT, 8><<<GRID, 512, 2 * GEMM_STAGE_BYTES, st>>>(ARGS, tiles_m, tiles_n);                  \
        else gemm_kernel<ACT, 4><<<GRID, 256, 2 * GEMM_STAGE_BYTES, st>>>(ARGS, tiles_m, tiles_n);                           \
    } while (0)
    if (S > 1) {
        float* ws = nullptr;
        int rc = splitk_workspace(st, (size_t)S * p->M * p->N * sizeof(float), &ws);
        if (rc) return rc;
        LinArgs b = a;                       // partial pass: raw fp32 accumulators into the slabs
        b.out_f32 = ws; b.ldo = p->N; b.out_bf16 = nullptr; b.bias = nullptr; b.residual = nullptr; b.act = SM_ACT_NONE;
        b.remap_in = 0; b.vt = nullptr;
        GEMM_LAUNCH(0, b, dim3(tiles, S));
        SM_LAUNCH_CHECK();
        if (ln_ok && S >= 2) {
            const PostLn ln = {p->post_ln_gamma, p->post_ln_beta, p->post_ln_eps, (bf16_t*)p->post_ln_out, p->post_ln_ldo, nullptr, SM_ACT_NONE};
            splitk_reduce_ln_kernel<4><<<cdiv(p->M, 4), 256, 0, st>>>(a, ws, S, p->N, ln);
            SM_LAUNCH_CHECK();
            *ln_done = true;
            return SM_OK;
        }
        static int ln_rows_max = -1;                 // SM_POST_LN_ROWS_MAX (default 1024; 256 = the round-5 rule, A/B)
        if (ln_rows_max < 0) { const char* e = getenv("SM_POST_LN_ROWS_MAX"); ln_rows_max = e ? atoi(e) : 1024; }
        // (round 6: up to 1024 rows -- a 512-stream decode step sums o_proj / down_proj's two slabs and norms the rows in ONE pass instead of a slab-sum launch
        //  + a norm launch: 11.7 + 10.5 us per product at 512 rows)
        if (p->post_ln_gamma && ln_fuse && p->M <= ln_rows_max && (p->N & 255) == 0 && p->N <= 4096 && (p->ldo & 3) == 0 && (!p->residual || (p->ldr & 3) == 0) &&
            p->remap_in == 0 && !a.wscale) {
            // few rows (a batched decode step of 33..128 streams): the row-block pass of the weight-streaming path -- slab sum, epilogue and the
            // LayerNorm / RMSNorm of the finished row in one launch (the slabs have the same [S][M][N] layout)
            const PostLn ln = {p->post_ln_gamma, p->post_ln_beta, p->post_ln_eps, (bf16_t*)p->post_ln_out, p->post_ln_ldo, p->post_ln_out_f32, p->post_ln_act};
            splitk_reduce_norm_rows_kernel<<<p->M, p->N / 4, 0, st>>>(a, ws, S, ln);
            SM_LAUNCH_CHECK();
            *ln_done = true;
            return SM_OK;
        }
        if (g_leave_slabs && !p->bias && !p->residual && p->act == SM_ACT_NONE && !a.wscale && !p->out_bf16 && !p->vt && p->remap_in == 0 && !p->post_ln_gamma) {
            g_leave_slabs->ws = ws; g_leave_slabs->S = S; g_leave_slabs->stride = (size_t)p->M * p->N;        // the caller's next kernel sums them on load
            return SM_OK;
        }
        const size_t nthr = (size_t)p->M * ((p->N + 3) / 4);
        splitk_reduce_kernel<<<(unsigned)((nthr + 255) / 256), 256, 0, st>>>(a, ws, nullptr, S, p->N);
        SM_LAUNCH_CHECK();
        return SM_OK;
    }
    const dim3 grid(tiles);
    if (p->act == SM_ACT_NONE) GEMM_LAUNCH(0, a, grid);
    else if (p->act == SM_ACT_QUICK_GELU) GEMM_LAUNCH(1, a, grid);
    else GEMM_LAUNCH(-1, a, grid);
#undef GEMM_LAUNCH
    SM_LAUNCH_CHECK();
    return SM_OK;
}
