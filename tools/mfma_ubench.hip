// Micro-benchmark: the compute phase of the 256x256 GEMM tile loop in isolation (LDS -> fragments -> MFMA, one
// barrier per 64-deep tile), for the two bf16 MFMA shapes.   hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
// variant with the staging traffic of the real kernel: every iteration each wave issues 8 x 1 KiB global_load_lds into the
// idle half of LDS (64 KiB per block per tile) from a big buffer, waits for the previous tile's loads, one barrier
__global__ __launch_bounds__(512, 2) void kload(const uint4* src, const char* big, size_t big_bytes, float* out, int iters, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 131072 / 16; i += 512) ((uint4*)smem)[i] = src[i];
    __syncthreads();
    const int wn = wave >> 2, wm = wave & 3;
    f32x4 acc[8][4];
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    const long long c0 = clock64(), w0 = wall_clock64();
    size_t off = ((size_t)blockIdx.x * 65536) % big_bytes;
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        char* dst = smem + ((it + 1) & 1) * 65536;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(big + off + (wave * 8 + j) * 1024 + lane * 16), (lds_ptr_t)(dst + (wave * 8 + j) * 1024), 16, 0, 0);
        off += (size_t)256 * 65536; if (off >= big_bytes) off -= big_bytes;
        const char* sw = smem + (it & 1) * 65536;
        const char* sx = sw + 32768;
        bf16x8 xf[2][4], wf[2][8];
#pragma unroll
        for (int ksl = 0; ksl < 2; ++ksl) {
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) xf[ksl][mf] = *(const bf16x8*)(sx + ((wm * 4 + mf) * 2 + ksl) * 1024 + lane * 16);
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) wf[ksl][nf] = *(const bf16x8*)(sw + ((wn * 8 + nf) * 2 + ksl) * 1024 + lane * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ksl = 0; ksl < 2; ++ksl)
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
                    acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ksl][nf], xf[ksl][mf], acc[nf][mf], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
    float t = 0;
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) t += acc[a][b][0] + acc[a][b][3];
    out[blockIdx.x * 512 + tid] = t;
}

template <int SHAPE, int BAR>
__global__ __launch_bounds__(512, 2) void k(const uint4* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 131072 / 16; i += 512) ((uint4*)smem)[i] = src[i];
    __syncthreads();
    const int wn = wave >> 2, wm = wave & 3;
    if (SHAPE == 16) {
        f32x4 acc[8][4];
        for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            const char* sw = smem + (it & 1) * 65536;
            const char* sx = sw + 32768;
            bf16x8 xf[2][4], wf[2][8];
#pragma unroll
            for (int ksl = 0; ksl < 2; ++ksl) {
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) xf[ksl][mf] = *(const bf16x8*)(sx + ((wm * 4 + mf) * 2 + ksl) * 1024 + lane * 16);
#pragma unroll
                for (int nf = 0; nf < 8; ++nf) wf[ksl][nf] = *(const bf16x8*)(sw + ((wn * 8 + nf) * 2 + ksl) * 1024 + lane * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ksl = 0; ksl < 2; ++ksl)
#pragma unroll
                for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf)
                        acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ksl][nf], xf[ksl][mf], acc[nf][mf], 0, 0, 0);
            if (BAR) __builtin_amdgcn_s_barrier();
        }
        float t = 0;
        for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) t += acc[a][b][0] + acc[a][b][3];
        out[blockIdx.x * 512 + tid] = t;
    } else {
        // 32x32x16: wave tile 128(n) x 64(m) = 4 x 2 fragments of 32x32, 4 k-slices of 16 per 64-deep tile
        f32x16 acc[4][2];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
        for (int it = 0; it < iters; ++it) {
            const char* sw = smem + (it & 1) * 65536;
            const char* sx = sw + 32768;
            bf16x8 xf[4][2], wf[4][4];
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) xf[kq][mf] = *(const bf16x8*)(sx + ((wm * 2 + mf) * 4 + kq) * 1024 + lane * 16);
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) wf[kq][nf] = *(const bf16x8*)(sw + ((wn * 4 + nf) * 4 + kq) * 1024 + lane * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf)
                        acc[nf][mf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kq][nf], xf[kq][mf], acc[nf][mf], 0, 0, 0);
            if (BAR) __builtin_amdgcn_s_barrier();
        }
        float t = 0;
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) t += acc[a][b][0] + acc[a][b][15];
        out[blockIdx.x * 512 + tid] = t;
    }
}

template <int SHAPE, int BAR>
void run(const uint4* src, float* out, const char* name) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute((const void*)k<SHAPE, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, BAR><<<blocks, 512, 131072>>>(src, out, 10);
    hipEventRecord(e0);
    k<SHAPE, BAR><<<blocks, 512, 131072>>>(src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 2.0 * 256 * 256 * 64 * iters * blocks;
    printf("%-28s %8.3f ms  %8.1f TFLOP/s   (%.3f us per 64-deep tile)\n", name, ms, flops / ms / 1e9, ms * 1e3 / iters);
}

int main() {
    uint4* src; float* out;
    hipMalloc(&src, 131072); hipMalloc(&out, 256 * 512 * 4);
    uint16_t* h = (uint16_t*)malloc(131072);
    srand(1);
    for (int i = 0; i < 65536; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f); uint32_t u; memcpy(&u, &f, 4); h[i] = u >> 16; }
    hipMemcpy(src, h, 131072, hipMemcpyHostToDevice);
    {
        const size_t big_bytes = (size_t)256 << 20;
        char* big; hipMalloc(&big, big_bytes + (1 << 20)); hipMemset(big, 0x3c, big_bytes + (1 << 20));
        long long* clk; hipMalloc(&clk, 256 * 16);
        hipFuncSetAttribute((const void*)kload, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        const int iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (size_t fp : {(size_t)16 << 20, (size_t)256 << 20}) {
            kload<<<256, 512, 131072>>>(src, big, fp, out, 10, clk);
            hipEventRecord(e0);
            kload<<<256, 512, 131072>>>(src, big, fp, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("16x16x32 + 64 KiB glds/tile, footprint %4zu MiB: %8.3f ms %8.1f TFLOP/s (%.3f us/tile), shader clock ~%.0f MHz\n",
                   fp >> 20, ms, 2.0 * 256 * 256 * 64 * iters * 256 / ms / 1e9, ms * 1e3 / iters, (double)h[0] / ((double)h[1] / 100.0));
        }
    }
    run<16, 1>(src, out, "16x16x32, barrier/tile");
    run<16, 0>(src, out, "16x16x32, no barrier");
    run<32, 1>(src, out, "32x32x16, barrier/tile");
    run<32, 0>(src, out, "32x32x16, no barrier");
    return 0;
}
