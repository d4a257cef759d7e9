// Probe: does an inline-asm v_mfma with a "+a" (AGPR-pinned) accumulator give the builtin's result?  One wave, 64 independent
// accumulator tiles updated over KSTEPS k-steps from LDS-read fragments (the access pattern of a 4-wave 128x128-per-wave GEMM).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_probe tools/mfma_asm_probe.hip && tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__device__ __forceinline__ void mma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    if (MODE == 0) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    } else {
        const u32x4 au = __builtin_bit_cast(u32x4, a), bu = __builtin_bit_cast(u32x4, b);
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(au), "v"(bu));
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void probe(const bf16x8* __restrict__ w, const bf16x8* __restrict__ x, float* __restrict__ out, int ksteps) {
    __shared__ bf16x8 lw[2][8][64], lx[2][8][64];
    const int lane = threadIdx.x;
    f32x4 acc[8][8];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0, 0, 0, 0};
    for (int q = 0; q < 8; ++q) { lw[0][q][lane] = w[q * 64 + lane]; lx[0][q][lane] = x[q * 64 + lane]; }
    __syncthreads();
    bf16x8 wa[8], xa[8], wb[8], xb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { wa[q] = lw[0][q][lane]; xa[q] = lx[0][q][lane]; }
    auto step = [&](int ks, bf16x8 (&wc)[8], bf16x8 (&xc)[8], bf16x8 (&wn)[8], bf16x8 (&xn)[8]) {
        // stage the next k-step's operands into the other LDS buffer (plain loads), read them into the other register set
        const int nb = (ks + 1) & 1;
        for (int q = 0; q < 8; ++q) {
            lw[nb][q][lane] = w[((size_t)(ks + 1) * 8 + q) * 64 + lane];
            lx[nb][q][lane] = x[((size_t)(ks + 1) * 8 + q) * 64 + lane];
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 16; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = g * 4 + e, m = t >> 3, n = t & 7;
                mma<MODE>(acc[n][m], wc[n], xc[m]);
            }
            if (g < 8) wn[g] = lw[nb][g][lane]; else xn[g - 8] = lx[nb][g - 8][lane];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int ks = 0; ks + 1 < ksteps; ks += 2) {
        step(ks, wa, xa, wb, xb);
        step(ks + 1, wb, xb, wa, xa);
    }
    if (MODE) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[((n * 8 + m) * 4 + r) * 64 + lane] = acc[n][m][r];
}

int main() {
    const int ksteps = 16;
    const size_t nfrag = (size_t)(ksteps + 1) * 8 * 64;
    std::vector<unsigned short> hw(nfrag * 8), hx(nfrag * 8);
    srand(1);
    for (auto& v : hw) v = 0x3c00 + (rand() & 0x1ff);          // bf16 around 0.008 .. 0.03
    for (auto& v : hx) v = 0xbc00 + (rand() & 0x1ff) + ((rand() & 1) << 15) * 0;
    void *dw, *dx; float *o0, *o1;
    hipMalloc(&dw, hw.size() * 2); hipMalloc(&dx, hx.size() * 2);
    hipMalloc(&o0, 64 * 4 * 64 * 4); hipMalloc(&o1, 64 * 4 * 64 * 4);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    probe<0><<<1, 64>>>((const bf16x8*)dw, (const bf16x8*)dx, o0, ksteps);
    probe<1><<<1, 64>>>((const bf16x8*)dw, (const bf16x8*)dx, o1, ksteps);
    std::vector<float> h0(64 * 4 * 64), h1(64 * 4 * 64);
    hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; double mx = 0;
    for (size_t i = 0; i < h0.size(); ++i) { if (h0[i] != h1[i]) ++bad; if (fabs(h0[i]) > mx) mx = fabs(h0[i]); }
    printf("asm vs builtin: %zu of %zu differ (max |ref| %.4g; sample %g vs %g)\n", bad, h0.size(), mx, h0[5], h1[5]);
    return bad ? 1 : 0;
}
