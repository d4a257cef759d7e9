// Do a wave's MFMAs and ANOTHER wave's VALU work on the same SIMD overlap?  One 8-wave block per CU (waves w and w + 4 share a
// SIMD).  Per "tile": 36 x v_mfma_f32_16x16x32_bf16 (the matrix phase of the ViT attention) and a softmax-like vector phase
// (32 v_exp_f32, 16 v_pk_fma_f32, 16 v_cvt_pk_bf16_f32, 16 v_max3_f32, 20 v_pk_mul_f32).
//   mode 0: every wave MFMA only            mode 1: every wave VALU only          mode 2: every wave MFMA then VALU (serial)
//   mode 3: waves 0-3 MFMA only, 4-7 VALU only (no barriers)
//   mode 4: two groups alternate M | V one barrier apart (the ping-pong schedule)       mode 5: like 4, s_setprio 1 around the MFMAs
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void mphase(f32x4 (&acc)[12], const bf16x8& a, const bf16x8& b) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int u = 0; u < 12; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
}
__device__ __forceinline__ void vphase(float (&s)[32], f32x2 (&o)[20], float c, float mc) {
    float mx = s[0];
#pragma unroll
    for (int u = 1; u < 31; u += 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(s[u]), "v"(s[u + 1]));
#pragma unroll
    for (int u = 0; u < 32; u += 2) {
        f32x2 t = f32x2{s[u], s[u + 1]} * f32x2{c, c} + f32x2{mc, mc};
        s[u] = __builtin_amdgcn_exp2f(t[0]);
        s[u + 1] = __builtin_amdgcn_exp2f(t[1]);
    }
    unsigned pk[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[u]) : "v"(s[2 * u]), "v"(s[2 * u + 1]));
#pragma unroll
    for (int u = 0; u < 20; ++u) o[u] *= f32x2{mx, mx};
#pragma unroll
    for (int u = 0; u < 16; ++u) s[2 * u] += __uint_as_float(pk[u] & 0xffff0000u) * 1e-30f;
}

__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = wave >> 2;
    f32x4 acc[12];
    for (int u = 0; u < 12; ++u) acc[u] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    float s[32];
    f32x2 o[20];
    for (int u = 0; u < 32; ++u) s[u] = 0.01f * (lane + u);
    for (int u = 0; u < 20; ++u) o[u] = f32x2{1.f, 1.f};
    __syncthreads();
    const long long t0 = clock64();
    if (mode == 0) for (int it = 0; it < iters; ++it) mphase(acc, a, b);
    else if (mode == 1) for (int it = 0; it < iters; ++it) vphase(s, o, 0.18f, -0.3f);
    else if (mode == 2) for (int it = 0; it < iters; ++it) { mphase(acc, a, b); vphase(s, o, 0.18f, -0.3f); }
    else if (mode == 3) { if (grp == 0) for (int it = 0; it < iters; ++it) mphase(acc, a, b); else for (int it = 0; it < iters; ++it) vphase(s, o, 0.18f, -0.3f); }
    else {
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int it = 0; it < iters; ++it) {
            if (mode == 5) __builtin_amdgcn_s_setprio(1);
            mphase(acc, a, b);
            if (mode == 5) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            vphase(s, o, 0.18f, -0.3f);
            __builtin_amdgcn_s_barrier();
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    float r = 0;
    for (int u = 0; u < 12; ++u) r += acc[u][0] + acc[u][3];
    for (int u = 0; u < 32; ++u) r += s[u];
    for (int u = 0; u < 20; ++u) r += o[u][0];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    for (int mode = 0; mode < 6; ++mode) {
        k<<<256, 512>>>(mode, 10, out, cyc);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<256, 512>>>(mode, iters, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int b = 0; b < 256; ++b) avg += h[b]; avg /= 256;
        printf("mode %d: %.1f us, %.0f clk per tile-iteration (per wave; a SIMD hosts 2 waves), %.2f GHz\n", mode, ms * 1e3, avg / iters, avg / (ms * 1e6));
    }
    return 0;
}
