// What does a device-side grid barrier cost on this part, next to the 4.7 us a dependent launch costs?
// Persistent grid (one block per CU), every iteration: a block publishes a value, all blocks meet at a barrier (one device-scope
// atomic arrive + bounded polling), every block reads ANOTHER block's value (which lives behind another XCD's L2) and checks it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/grid_barrier_ubench tools/experiments/grid_barrier_fenced_ubench.hip   (the round-3 file grid_barrier_ubench.hip is the same barrier WITHOUT the fences: 3.7 us, but nothing is published)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>   // 0: flat counter; 1: per-XCD counters (block b on XCD b % 8) + a top-level counter
__global__ __launch_bounds__(512) void barrier_kernel(unsigned* counters, int iters, float* data, int* errors, long long timeout) {
    const int nb = gridDim.x, b = blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0) __hip_atomic_store(&data[b], (float)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            if (MODE == 0) {
                __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(it + 1) * nb;
                while (__hip_atomic_load(&counters[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (wall_clock64() - t0 > timeout) { atomicAdd(errors, 1); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            } else {
                const int x = b & 7, per = (nb + 7 - x) / 8;             // blocks on this XCD
                const unsigned prev = __hip_atomic_fetch_add(&counters[64 * (1 + x)], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (prev == (unsigned)(it + 1) * per - 1) __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(it + 1) * 8;
                while (__hip_atomic_load(&counters[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (wall_clock64() - t0 > timeout) { atomicAdd(errors, 1); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float v = __hip_atomic_load(&data[(b + 37) % nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v < (float)(it + 1)) atomicAdd(errors + 1, 1);
        }
    }
}

__global__ void empty_kernel(float* d) { if (threadIdx.x == 0 && blockIdx.x == 9999) d[0] = 1.f; }

int main() {
    unsigned* counters; float* data; int* errors;
    CK(hipMalloc(&counters, 64 * 9 * sizeof(unsigned) * 4)); CK(hipMalloc(&data, 4096 * sizeof(float))); CK(hipMalloc(&errors, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long timeout = 2LL * 100000000;    // 2 s of wall_clock64 (100 MHz)
    for (int threads : {256, 512}) for (int nb : {64, 256, 512}) for (int mode = 0; mode < 2; ++mode) {
        const int iters = 2000;
        CK(hipMemset(counters, 0, 64 * 9 * sizeof(unsigned) * 4)); CK(hipMemset(data, 0, 4096 * 4)); CK(hipMemset(errors, 0, 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (mode == 0) barrier_kernel<0><<<nb, threads>>>(counters, iters, data, errors, timeout);
        else barrier_kernel<1><<<nb, threads>>>(counters, iters, data, errors, timeout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int err[2]; CK(hipMemcpy(err, errors, 8, hipMemcpyDeviceToHost));
        printf("blocks %3d x %3d threads, %s counter: %.2f us per barrier (timeouts %d, stale reads %d)\n", nb, threads, mode ? "per-XCD + top" : "flat", ms * 1e3 / iters, err[0], err[1]);
    }
    // the launch floor on the same box: 2000 dependent empty launches
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 2000; ++i) empty_kernel<<<4, 256>>>(data);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("dependent empty launches: %.2f us each\n", ms * 1e3 / 2000);
    return 0;
}
