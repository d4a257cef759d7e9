// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef uint16_t bf16_t;   // raw storage type in HBM

#define SM_WAVE 64

// 16-byte WRITE-THROUGH store (global_store_dwordx4 ... sc1): the line leaves the XCD's L2 as it is written instead of
// staying dirty until the end-of-kernel L2 write-back.  Big outputs that the NEXT kernel reads (GEMM / LayerNorm / attention
// results) use it: with plain stores every kernel boundary paid dirty-bytes / ~6 TB/s (5.5-7 us behind a ViT-batch GEMM).
// The trailing s_nop is part of the contract: a VMEM store of more than 64 bits reads its data registers over more than one cycle,
// and a VALU write of those registers in the next wait states corrupts the tail of the data (ISA "manually inserted wait states").
// hipcc pads that hazard for stores it can see -- not inside inline asm: the persistent GEMM's epilogue (address arithmetic of the
// next store allocated into the data registers of the previous one) wrote an address word into 1 of 40 000 outputs without it.
__device__ __forceinline__ void store16_wt(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// The same store for an output that is STREAMED -- written once, read once by the next kernel -- and large enough to push the residual stream out of the
// 256 MB Infinity Cache: fc1's activation of a 28-frame lane is 132 MB, written between out-proj's update of the 66 MB fp32 residual and fc2's read-modify-write
// of it.  With `nt` (no allocation on the way through) the residual is still there when fc2's epilogue asks: measured in situ, same box, per launch
// (profiles/r06_nt_store_ab.txt): fc2 123.4 -> 113.5 us (its residual read comes from the cache: 126 -> 113 us is also what fc2 costs in isolation, where the
// residual never leaves it), fc1 116.5 -> 121.7 us (the stores go out to HBM instead of being absorbed); frames/s +0.5-1 % with two lanes, -0.3..+0.9 % single
// lane depending on the box.  The q|k|v output the same way loses (+9.5 us on q|k|v, out-proj unchanged: its residual was resident anyway); plain `nt` (no sc1) and
// `sc0 sc1 nt` measure like `sc1 nt` on fc2 and worse / equal on fc1.  SM_STREAM_STORE_MODS picks the bits for A/B builds (tools/build_variant.sh, tools/nt_store_ab.sh).
#ifndef SM_STREAM_STORE_MODS
#define SM_STREAM_STORE_MODS "sc1 nt"
#endif
__device__ __forceinline__ void store16_stream(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off " SM_STREAM_STORE_MODS "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// round-to-nearest-even fp32 -> bf16 bits (hardware v_cvt_pk_bf16_f32; NaN stays NaN); matches torch .to(bfloat16)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ uint32_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return __builtin_bit_cast(uint32_t, bf16x2{(__bf16)lo, (__bf16)hi}); }

// ---- 16-bit operand type of the MFMA products: bf16 (default everywhere) or IEEE fp16 (opt-in for the ViT: the reference's own
// demo precision, model/builder.py:54 -- same MFMA rate, 3 more mantissa bits on every activation, fp16 range suffices for the
// tower's LayerNorm outputs / attention context / MLP activations).  Storage stays raw uint16 / `bf16x8`-typed 16-byte
// fragments (LDS-DMA, layouts and swizzles are type-agnostic); only the MFMA opcode and the fp32 <-> 16-bit conversions differ.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
typedef __attribute__((ext_vector_type(16))) float f32x16;
// 32x32x16: the same 16 operand bytes per lane as 16x16x32 and twice the work per issued instruction (32 pipe cycles per SIMD,
// back to back at exactly that rate; the 16x16x32 form issues at 17-18 cycles for its nominal 16).  A: lane l holds
// A[l % 32][8 * (l / 32) .. +8]; B likewise; D: register j of lane l is D[8 * (j / 4) + 4 * (l / 32) + j % 4][l % 32].
template <bool F16>
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }      // round to nearest even
__device__ __forceinline__ float h2f(uint32_t b) { return (float)__builtin_bit_cast(_Float16, (uint16_t)b); }
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) { return __builtin_bit_cast(uint32_t, f16x2{(_Float16)lo, (_Float16)hi}); }
template <bool F16> __device__ __forceinline__ uint32_t cvt16(float f) { if constexpr (F16) return f2h(f); else return f2bf(f); }
template <bool F16> __device__ __forceinline__ uint32_t pack16(float lo, float hi) { if constexpr (F16) return pack2h(lo, hi); else return pack2bf(lo, hi); }
template <bool F16> __device__ __forceinline__ float up16(uint32_t b) { if constexpr (F16) return h2f(b); else return bf2f(b); }
__device__ __forceinline__ uint32_t cvt16_rt(float f, int f16) { return f16 ? f2h(f) : f2bf(f); }
__device__ __forceinline__ uint32_t pack16_rt(float lo, float hi, int f16) { return f16 ? pack2h(lo, hi) : pack2bf(lo, hi); }

// cross-row exchanges on the VALU (v_permlane16_swap / v_permlane32_swap, gfx950) instead of ds_bpermute's LDS round trip:
// swapping x with itself leaves {row0,row0,row2,row2} / {row1,row1,row3,row3} (resp. the two 32-lane halves) in the pair
__device__ __forceinline__ float xor16_max(float x) {
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float x) {
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_sum(float x) {
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float x) {
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// LayerNorm of one row held by one wave (D = 256 * NV columns, lane owns columns lane*4 + j*256) for the fused slab-sum + LayerNorm
// pass of the split-K GEMM (linear.hip): mean, then the sum of squared deviations, in registers -- norm_wave_fixed_kernel's algorithm
// with the operation sequence pinned (fused multiply-adds written out, nothing left to contraction).  The standalone kernel keeps
// the code it was validated with in round 3, so the two may differ in the last fp32 bit of an intermediate (a handful of 16-bit
// outputs one ulp apart: test_linear_post_ln states the bound).
template <int NV>
__device__ __forceinline__ void ln_row_stats(const f32x4 (&v)[NV], float eps, float& mu, float& rstd) {
#pragma clang fp contract(off)
    constexpr float inv_d = 1.0f / (NV * 256);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    s = wave_sum(s);
    mu = s * inv_d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; q = __builtin_fmaf(d, d, q); }
    q = wave_sum(q);
    rstd = rsqrtf(__builtin_fmaf(q, inv_d, eps));
}
__device__ __forceinline__ float ln_apply(float v, float mu, float rstd, float g, float b) {
#pragma clang fp contract(off)
    const float t = (v - mu) * rstd;
    return __builtin_fmaf(t, g, b);
}

// 1/(1+e^-x) on the transcendental unit: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the ~20-instruction IEEE division;
// x -> -inf gives rcp(inf) = 0, x -> +inf gives rcp(1) = 1
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }

// activations applied in GEMM epilogues (include/streammind_hip.h SM_ACT_*)
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case 1: return v * sigmoidf_(1.702f * v);                         // quick_gelu (HF CLIP)
        case 2: return v >= 0.f ? v : 0.01f * v;                          // leaky_relu, slope 0.01
        case 3: return v > 20.f ? v : log1pf(__expf(v));                  // softplus (beta 1, threshold 20)
        case 4: return siluf_(v);
        case 5: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));   // nn.GELU() (exact, erf): STC readout builder.py:566-571
        default: return v;
    }
}

// Packed ("fragment-major") weight layout used by every linear op:
//   W[N][K] row-major bf16  ->  Wp[N/16][K/32][lane 0..63][8]   (N, K zero-padded to 16 / 32)
//   lane = g*16 + i holds W[rg*16 + i][ks*32 + g*8 + 0..7]
// so one wave-wide 16-byte load of (rg, ks) is exactly the A operand of
// v_mfma_f32_16x16x32_bf16 and is 1 KiB contiguous in HBM and conflict-free in LDS.
__host__ __device__ __forceinline__ size_t packed_index(int n, int k, int KS) {
    return ((size_t)(n >> 4) * KS + (k >> 5)) * 512 + (size_t)((((k & 31) >> 3) << 4) + (n & 15)) * 8 + (k & 7);
}
