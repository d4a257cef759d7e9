// Which R | M interval length suits the 256x256 GEMM block?  8 waves in two groups one barrier apart (the schedule of
// gemm256.hip), fragments from LDS, optional LDS-DMA traffic of the real kernel (4 pieces of 1 KiB per wave per 32-deep k-step,
// scalar-base form), 256 blocks = one per CU.
//   STEP = 32: R = 12 fragment reads (+ 4 DMA pieces), M = 32 MFMAs, two barriers per 32-deep k-step     (gemm256.hip today)
//   STEP = 64: R = 24 fragment reads (+ 8 DMA pieces), M = 64 MFMAs, two barriers per 64-deep k-step     (96 fragment registers)
// Measured (constant operand data, so the chip clocks ~2.3 GHz; us per 64-deep tile per CU): LDS only 1.004 (STEP 32) / 0.984
// (STEP 64): the barriers cost 2 %.  With the staging traffic from a 64 MiB window 1.39-1.43 / 1.60-1.80: the 64 KiB per tile
// that every CU pulls through L2 -> LDS (11.7 TB/s chip-wide against ~15 measured peak) is what stretches the loop, and bunching
// 8 pieces into one R interval makes it worse.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/gemm_sched_ubench.hip -o /tmp/gs && WIN_MB=64 /tmp/gs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int STEP, bool DMA>
__global__ __launch_bounds__(512, 2) void k(const uint4* src, const char* big, size_t big_bytes, float* out, int tiles64) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 131072 / 16; i += 512) ((uint4*)smem)[i] = src[i];
    __syncthreads();
    const int wn = wave >> 2, wm = wave & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const uint32_t voff = lane * 16;
    f32x4 acc[8][4];
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    // source window of big_bytes (>= 16 MiB), walked in 32-KiB strides per block: L2 / Infinity-Cache resident like the real operands
    size_t goff = ((size_t)blockIdx.x * 32768) % (big_bytes - 65536);
    const char* gsrc = big + goff + (size_t)wave * 4096;
    constexpr int H = STEP / 32;                   // 32-deep half steps per interval pair
    const int steps = tiles64 * 2 / H;
    if (wn == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < steps; ++t) {
        bf16x8 xf[H][4], wf[H][8];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int slot = (t * H + h) & 3;
            const char* sw = smem + slot * 32768;
            const char* sx = sw + 16384;
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) xf[h][mf] = *(const bf16x8*)(sx + (wm * 4 + mf) * 1024 + lane * 16);
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) wf[h][nf] = *(const bf16x8*)(sw + (wn * 8 + nf) * 1024 + lane * 16);
        }
        if (DMA) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const int slot = (t * H + h + 3) & 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t dst = lds0 + slot * 32768 + (j >> 1) * 16384 + (wave * 2 + (j & 1)) * 1024;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gsrc + j * 1024), "s"(dst) : "memory");
                }
                goff = (goff + (size_t)256 * 32768) % (big_bytes - 65536);
                gsrc = big + goff + (size_t)wave * 4096;
            }
            if (H == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
                    acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][nf], xf[h][mf], acc[nf][mf], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wn == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = 0;
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) r += acc[a][b][0] + acc[a][b][3];
    out[blockIdx.x * 512 + tid] = r;
}
template <int STEP, bool DMA> void run(const uint4* src, const char* big, size_t bb, float* out, const char* name) {
    const int tiles = 2000, blocks = 256;
    hipFuncSetAttribute((const void*)k<STEP, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<STEP, DMA><<<blocks, 512, 131072>>>(src, big, bb, out, 10);
    hipEventRecord(e0);
    k<STEP, DMA><<<blocks, 512, 131072>>>(src, big, bb, out, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 256 * 256 * 64 * tiles * blocks;
    printf("%-34s %8.3f ms  %8.1f TFLOP/s   (%.3f us per 64-deep tile)\n", name, ms, flops / ms / 1e9, ms * 1e3 / tiles);
}
int main() {
    uint4* src; float* out; char* big;
    int win = getenv("WIN_MB") ? atoi(getenv("WIN_MB")) : 64;
    if (win < 16) win = 16;
    const size_t bb = (size_t)win << 20;
    hipMalloc(&src, 131072); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&big, bb);
    hipMemset(src, 0x3c, 131072); hipMemset(big, 0x3c, bb);
    run<32, false>(src, big, bb, out, "STEP 32, LDS only");
    run<64, false>(src, big, bb, out, "STEP 64, LDS only");
    run<32, true>(src, big, bb, out, "STEP 32, with LDS-DMA traffic");
    run<64, true>(src, big, bb, out, "STEP 64, with LDS-DMA traffic");
    run<32, true>(src, big, bb, out, "STEP 32, with LDS-DMA traffic");
    run<64, true>(src, big, bb, out, "STEP 64, with LDS-DMA traffic");
    return 0;
}
