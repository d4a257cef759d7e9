// Which CUs does a CU-masked HIP stream use on this part?  (hipExtStreamCreateWithCUMask; prerequisite of running the two tower lanes
// on disjoint halves of the chip.)  hipcc --offload-arch=gfx950 -O2 cu_mask_census.hip -o /tmp/cu_mask_census && /tmp/cu_mask_census
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
__global__ void census(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
        out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
    }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}
static void run(const char* name, hipStream_t st) {
    const int n = 4096;
    unsigned* d; hipMalloc(&d, n * 4);
    census<<<n, 64, 0, st>>>(d, 2000);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus; int per_xcc[16] = {0};
    std::set<unsigned> pxs[16];
    for (unsigned v : h) { const unsigned key = (v >> 16) << 16 | ((v >> 13) & 7) << 8 | ((v >> 8) & 0xf) | ((v >> 12) & 1) << 12; cus.insert(key); pxs[v >> 16].insert(key); }
    printf("%-22s distinct CUs %3zu; per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %zu", pxs[x].size());
    printf("\n");
    hipFree(d);
}
int main() {
    hipStream_t s0; hipStreamCreate(&s0); run("unmasked", s0);
    for (int variant = 0; variant < 3; ++variant) {
        uint32_t lo[8], hi[8];
        for (int w = 0; w < 8; ++w) {
            if (variant == 0) { lo[w] = w < 4 ? 0xffffffffu : 0; hi[w] = w < 4 ? 0 : 0xffffffffu; }          // bits 0..127 | 128..255
            else if (variant == 1) { lo[w] = 0x0000ffffu; hi[w] = 0xffff0000u; }                           // alternating 16-bit groups
            else { lo[w] = 0x55555555u; hi[w] = 0xaaaaaaaau; }                                             // even | odd bits
        }
        hipStream_t a, b;
        hipError_t e1 = hipExtStreamCreateWithCUMask(&a, 8, lo), e2 = hipExtStreamCreateWithCUMask(&b, 8, hi);
        if (e1 != hipSuccess || e2 != hipSuccess) { printf("variant %d: create failed %d %d\n", variant, e1, e2); continue; }
        char nm[64];
        snprintf(nm, sizeof nm, "variant %d mask A", variant); run(nm, a);
        snprintf(nm, sizeof nm, "variant %d mask B", variant); run(nm, b);
    }
    return 0;
}
