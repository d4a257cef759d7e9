// Round 5: the device-side grid barrier measured in the form MI355X_MICROARCH.md's price table prices ("barrier-xcd", "barrier-counter"),
// next to round 4's slow form (grid_barrier_fenced_ubench.hip: every thread fences, every waiter polls one counter with ACQUIRE loads).
//   barrier-counter : one monotonic counter; lane 0: release fence -> asm vmcnt(0) -> relaxed arrive; relaxed sc1 polling + s_sleep;
//                     ONE acquire fence after the wait.
//   barrier-xcd     : per-XCC counter (the block's XCC id is read from the hardware, the per-XCC block counts come from a census at the
//                     kernel's start: nothing depends on a placement rule); blocks drain their own stores (vmcnt(0)) and arrive relaxed on
//                     their XCC's counter; the XCC's last arriver does ONE release fence for the whole XCD L2, arrives on the top counter,
//                     polls it, acquires, and bumps the XCC's generation word; everybody else polls that word relaxed and acquires once.
// Payloads: nothing / a 128-byte record per block / a 16 KiB fp32 slab per block (plain stores), and every block then READS another
// block's payload (another XCD: b + 37) with plain loads and checks every word (stale reads counted).
// Also: the GPU-side cost of a dependent kernel boundary, as the slope of a hipGraph replay of N vs 2N trivial kernels (a host loop
// of launches measures the host's issue rate instead).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbx tools/experiments/grid_barrier_xcd_ubench.hip && /tmp/gbx
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Sync {
    unsigned top;            unsigned pad0[31];
    unsigned xcc_cnt[8][32];          // one 128-byte line per XCC
    unsigned xcc_gen[8][32];
    unsigned census[8];      unsigned pad1[24];
    unsigned flat;           unsigned pad2[31];
    unsigned timeouts, stale;
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7); }     // HW_REG_XCC_ID, low bits

// bounded relaxed poll: *p >= target
__device__ __forceinline__ void poll_ge(const unsigned* p, unsigned target, Sync* s) {
    const long long t0 = wall_clock64();
    while (ld_relaxed(p) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 200000000LL) { atomicAdd(&s->timeouts, 1u); break; }
    }
}

// MODE 0: barrier-counter, 1: barrier-xcd.  gen = 1, 2, ... ; per = blocks on this block's XCC
template <int MODE>
__device__ __forceinline__ void grid_barrier(Sync* s, unsigned gen, int xcc, unsigned per, unsigned nb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have been acknowledged by L2
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&s->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            poll_ge(&s->flat, gen * nb, s);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            const unsigned prev = __hip_atomic_fetch_add(&s->xcc_cnt[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gen * per - 1) {                       // this XCC's last arriver: one write-back for the whole XCD L2
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                poll_ge(&s->top, gen * 8u, s);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(&s->xcc_gen[xcc][0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                poll_ge(&s->xcc_gen[xcc][0], gen, s);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
    }
    __syncthreads();
}

// PAY: floats per block published per iteration (0, 32 = 128 B, 4096 = 16 KiB)
template <int MODE, int PAY>
__global__ __launch_bounds__(512) void barrier_kernel(Sync* s, int iters, float* data, int n_xcc_expected) {
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const int xcc = xcc_id();
    // census: how many blocks sit on my XCC (one flat barrier, once)
    if (threadIdx.x == 0) {
        atomicAdd(&s->census[xcc], 1u);
        __hip_atomic_fetch_add(&s->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        poll_ge(&s->flat, nb, s);
    }
    __syncthreads();
    const unsigned per = ld_relaxed(&s->census[xcc]);
    unsigned gen0 = 0;
    if (MODE == 0) gen0 = 1;            // the census already used generation 1 of the flat counter
    float* mine = data + (size_t)b * (PAY ? PAY : 1);
    const float* other = data + (size_t)((b + 37) % nb) * (PAY ? PAY : 1);
    for (int it = 0; it < iters; ++it) {
        const float tag = (float)(it + 1);
        if (PAY) for (int i = threadIdx.x; i < PAY; i += blockDim.x) mine[i] = tag + (float)i * 0.0f;
        grid_barrier<MODE>(s, gen0 + (unsigned)it * 2 + 1, xcc, per, nb);
        if (PAY) {
            int bad = 0;
            for (int i = threadIdx.x; i < PAY; i += blockDim.x) bad += other[i] != tag;
            if (bad) atomicAdd(&s->stale, (unsigned)bad);
            // second barrier: nobody overwrites a payload that is still being read
        }
        grid_barrier<MODE>(s, gen0 + (unsigned)it * 2 + 2, xcc, per, nb);
    }
}

__global__ void trivial_kernel(float* d) { if (threadIdx.x == 0 && blockIdx.x == 99999) d[0] = 1.f; }
// a "real" small streaming kernel: 256 blocks each touch 16 KiB
__global__ void touch_kernel(float* d) { d[(size_t)blockIdx.x * 4096 + threadIdx.x * 16] += 1.f; }

template <int MODE, int PAY>
static void run(Sync* s, float* data, int nb, int threads, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 1000;
    CK(hipMemset(s, 0, sizeof(Sync)));
    CK(hipMemset(data, 0, (size_t)1024 * 4096 * 4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    barrier_kernel<MODE, PAY><<<nb, threads>>>(s, iters, data, 8);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    Sync h; CK(hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost));
    printf("%-15s %4d blocks x %3d threads, payload %5d B/block: %6.2f us per barrier (2 per iteration; timeouts %u, stale words %u; census %u %u %u %u %u %u %u %u)\n",
           MODE ? "barrier-xcd" : "barrier-counter", nb, threads, PAY * 4, ms * 1e3 / (2 * iters), h.timeouts, h.stale,
           h.census[0], h.census[1], h.census[2], h.census[3], h.census[4], h.census[5], h.census[6], h.census[7]);
}

template <class K>
static float graph_time(K kern, int grid, int block, float* data, int n, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) kern<<<grid, block, 0, st>>>(data);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

int main() {
    Sync* s; float* data;
    CK(hipMalloc(&s, sizeof(Sync))); CK(hipMalloc(&data, (size_t)1024 * 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int threads : {256, 512}) for (int nb : {256, 512}) {
        if (nb == 512 && threads == 512) continue;                     // 2 x 512 threads per CU: fine, but keep the table short
        run<0, 0>(s, data, nb, threads, e0, e1);
        run<1, 0>(s, data, nb, threads, e0, e1);
        run<0, 32>(s, data, nb, threads, e0, e1);
        run<1, 32>(s, data, nb, threads, e0, e1);
        run<0, 4096>(s, data, nb, threads, e0, e1);
        run<1, 4096>(s, data, nb, threads, e0, e1);
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; ++rep) {
        const float a = graph_time(trivial_kernel, 256, 256, data, 200, st, e0, e1), b = graph_time(trivial_kernel, 256, 256, data, 400, st, e0, e1);
        printf("graph replay, trivial 256-block kernels: 200 -> %.1f us, 400 -> %.1f us: %.2f us per dependent kernel (slope)\n", a * 1e3, b * 1e3, (b - a) * 1e3 / 200);
        const float c = graph_time(touch_kernel, 256, 256, data, 200, st, e0, e1), d = graph_time(touch_kernel, 256, 256, data, 400, st, e0, e1);
        printf("graph replay, 256-block kernels touching 16 KiB each: 200 -> %.1f us, 400 -> %.1f us: %.2f us per dependent kernel (slope)\n", c * 1e3, d * 1e3, (d - c) * 1e3 / 200);
    }
    // host loop of dependent launches on the same box (round 4's "launch floor")
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 2000; ++i) trivial_kernel<<<4, 256>>>(data);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("host loop of 2000 dependent trivial launches: %.2f us each\n", ms * 1e3 / 2000);
    return 0;
}
