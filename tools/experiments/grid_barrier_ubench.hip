// What does a grid-wide barrier cost on MI355X (8 XCDs, one block per CU), built from agent-scope atomics with no cache-wide
// fence?  Needed to price a persistent decode-layer kernel (six dependent phases per layer).  Spins are bounded: a block that waits
// more than ~0.1 s sets an error flag and leaves.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/grid_barrier_ubench.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1L << 22)) { *err = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}
__global__ __launch_bounds__(512) void k(unsigned* counter, int n, int* err, float* out) {
    float acc = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        if (!grid_barrier(counter, (unsigned)(i + 1) * gridDim.x, err)) break;
        acc = acc * 1.0001f + 1.f;
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
int main() {
    unsigned* counter; int* err; float* out;
    hipMalloc(&counter, 4); hipMalloc(&err, 4); hipMalloc(&out, 512 * 512 * 4);
    for (int blocks : {64, 128, 256}) {
        for (int n : {10, 2000}) {
            hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            k<<<blocks, 512>>>(counter, n, err, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int h; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
            if (n > 10) printf("%d blocks: %.2f us per grid barrier (err %d)\n", blocks, ms * 1e3 / n, h);
        }
    }
    return 0;
}
