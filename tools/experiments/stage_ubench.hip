// Per-CU staging rate of L2/MALL-resident data as a function of the number of active CUs, for the two ways a GEMM block can fetch an
// operand: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) and plain 16-byte loads into VGPRs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stage_ubench tools/experiments/stage_ubench.hip && /tmp/stage_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every wave: `iters` rounds of DEPTH pieces in flight; window bytes are re-read (L2-resident when small)
template <int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const char* src, size_t window, int iters, unsigned* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* my = smem + wave * (DEPTH * 1024);
    size_t off = ((size_t)blockIdx.x * 8 + wave) * 65536 % window;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off + lane * 16), (lds_ptr_t)(my + d * 1024), 16, 0, 0);
            off += 1024; if (off >= window) off = 0;
        }
        if (DEPTH >= 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && iters < 0) sink[0] = *(unsigned*)smem;
}
template <int DEPTH>
__global__ __launch_bounds__(512) void vgpr_kernel(const char* src, size_t window, int iters, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t off = ((size_t)blockIdx.x * 8 + wave) * 65536 % window;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            v[d] = __builtin_nontemporal_load((const u32x4*)(src + off + lane * 16));
            off += 1024; if (off >= window) off = 0;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

int main() {
    const size_t cap = 512ull << 20;
    char* src; unsigned* sink;
    CK(hipMalloc(&src, cap)); CK(hipMalloc(&sink, 64)); CK(hipMemset(src, 1, cap));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)dma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)dma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const size_t windows[] = {2ull << 20, 24ull << 20, 512ull << 20};       // L2-resident / MALL-resident / HBM
    const int blocks[] = {16, 40, 80, 120, 160, 256, 512};
    for (size_t w : windows)
        for (int nb : blocks) {
            const int iters = 512;           // x 8 pieces x 8 waves = 32 MiB per block
            float ms[3];
            for (int kind = 0; kind < 3; ++kind) {
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    if (kind == 0) dma_kernel<8><<<nb, 512, 65536>>>(src, w, iters, sink);
                    else if (kind == 1) dma_kernel<4><<<nb, 512, 32768>>>(src, w, iters * 2, sink);
                    else vgpr_kernel<8><<<nb, 512>>>(src, w, iters, sink);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms[kind], e0, e1));
                }
            }
            const double bytes = (double)iters * 8 * 8 * 1024;
            printf("window %4zu MiB blocks %3d : per-block KB/us  dma8 %6.1f  dma4 %6.1f  vgpr8 %6.1f   | chip TB/s dma8 %5.2f vgpr8 %5.2f\n", w >> 20, nb,
                   bytes / (ms[0] * 1e3) / 1e3, bytes / (ms[1] * 1e3) / 1e3, bytes / (ms[2] * 1e3) / 1e3, bytes * nb / (ms[0] * 1e-3) / 1e12, bytes * nb / (ms[2] * 1e-3) / 1e12);
        }
    return 0;
}
