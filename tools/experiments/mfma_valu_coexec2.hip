// Which VALU instructions overlap with MFMAs, and from where?  8-wave block per CU (waves w, w + 4 share a SIMD).
//   A: all waves 36 MFMAs per iteration            B: all waves 128 x OP per iteration
//   X: waves 0-3 MFMA only, waves 4-7 OP only (cross-wave overlap?)
//   I: every wave: MFMA followed by FILL x OP, 36 times (intra-wave overlap; FILL = 3)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/mfma_valu_coexec2.hip -o /tmp/coexec2 && /tmp/coexec2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int OP> __device__ __forceinline__ void vop(float& x, f32x2& y) {
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
    if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(y));
    if (OP == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x));
    if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x));
    if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(y));
    if (OP == 6) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
}
template <int OP, int MODE>
__global__ __launch_bounds__(512) void k(int iters, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = wave >> 2;
    f32x4 acc[12];
    for (int u = 0; u < 12; ++u) acc[u] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    float x[8]; f32x2 y[8];
    for (int u = 0; u < 8; ++u) { x[u] = 0.001f * (lane + u); y[u] = f32x2{x[u], x[u]}; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 2 && grp == 0)) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int u = 0; u < 12; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
        }
        if (MODE == 1 || (MODE == 2 && grp == 1)) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u) vop<OP>(x[u], y[u]);
        }
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int f = 0; f < 3; ++f) vop<OP>(x[(u * 3 + f) & 7], y[(u * 3 + f) & 7]);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    float r = 0;
    for (int u = 0; u < 12; ++u) r += acc[u][0] + acc[u][3];
    for (int u = 0; u < 8; ++u) r += x[u] + y[u][0] + y[u][1];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int OP, int MODE> float run(float* out) {
    const int iters = 1000;
    k<OP, MODE><<<256, 512>>>(10, out);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<OP, MODE><<<256, 512>>>(iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters * 1e3f;      // ns per iteration
}
template <int OP> void row(const char* name, float* out, float tA) {
    const float tB = run<OP, 1>(out), tX = run<OP, 2>(out), tI = run<OP, 3>(out);
    printf("%-18s B(all waves 128 op) %6.0f ns | X(4 waves MFMA + 4 waves 128 op) %6.0f ns (sum of halves %6.0f, max %6.0f) | I(each wave 36 x (MFMA + 3 op)) %6.0f ns (A %6.0f + 108 ops %6.0f)\n",
           name, tB, tX, tA / 2 + tB / 2, tA / 2 > tB / 2 ? tA / 2 : tB / 2, tI, tA, tB * 108 / 128);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const float tA = run<0, 0>(out);
    printf("A: all waves 36 MFMA / iteration: %.0f ns (2 waves per SIMD)\n", tA);
    row<0>("v_fma_f32", out, tA); row<6>("v_add_f32", out, tA); row<1>("v_exp_f32", out, tA); row<2>("v_pk_fma_f32", out, tA);
    row<5>("v_pk_mul_f32", out, tA); row<3>("v_max3_f32", out, tA); row<4>("v_cvt_pk_bf16_f32", out, tA);
    return 0;
}
