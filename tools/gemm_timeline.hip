// Per-block phase timeline of the 256x256 tiled GEMM (prologue / main loop / epilogue halves) on the ViT shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSM_GEMM_TIMELINE -Istreammind_amd/csrc -Iinclude tools/gemm_timeline.hip -o /tmp/gtl && /tmp/gtl [M]
// Operand VALUES are random bf16 written straight into the packed layout (timing does not care which matrix it is).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../streammind_amd/csrc/gemm256.hip"
thread_local char g_sm_err[512];

static void fill_bf16(void* d, size_t n, float scale) {
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2 * scale; uint32_t u; memcpy(&u, &f, 4); h[i] = u >> 16; }
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16156;
    struct { const char* name; int N, K, act, resid, obf; } shapes[] = {
        {"qkv", 3072, 1024, 0, 0, 1}, {"out", 1024, 1024, 0, 1, 0}, {"fc1", 4096, 1024, 1, 0, 1}, {"fc2", 1024, 4096, 0, 1, 0}};
    long long* tl; hipMalloc(&tl, 4096 * 8 * 8 + 256 * 8 * 4 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_timeline), &tl, sizeof(tl));
    for (auto& sh : shapes) {
        void *w, *x, *ob; float *bias, *res, *of;
        hipMalloc(&w, (size_t)sh.N * sh.K * 2); hipMalloc(&x, (size_t)M * sh.K * 2);
        hipMalloc(&ob, (size_t)M * sh.N * 2); hipMalloc(&of, (size_t)M * sh.N * 4); hipMalloc(&res, (size_t)M * sh.N * 4);
        hipMalloc(&bias, sh.N * 4); hipMemset(bias, 0, sh.N * 4); hipMemset(res, 0, (size_t)M * sh.N * 4);
        fill_bf16(w, (size_t)sh.N * sh.K, 0.03f); fill_bf16(x, (size_t)M * sh.K, 1.0f);
        LinArgs a; memset(&a, 0, sizeof(a));
        a.w = (const bf16x8*)w; a.N = sh.N; a.K = sh.K; a.KS = sh.K / 32; a.NRG = sh.N / 16; a.x = x; a.M = M; a.ldx = sh.K;
        a.bias = bias; a.act = sh.act; a.residual = sh.resid ? res : nullptr; a.ldr = sh.N;
        a.out_f32 = sh.obf ? nullptr : of; a.out_bf16 = sh.obf ? (bf16_t*)ob : nullptr; a.ldo = sh.N; a.ldo_bf16 = sh.N;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it = 0; it < 3; ++it) launch_gemm256(a, sh.act, 256, 0);
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) launch_gemm256(a, sh.act, 256, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int nblk = ((M + 255) / 256) * (sh.N / 256);
        std::vector<long long> h((size_t)nblk * 8);
        hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost);
        long long t0 = h[0], t1 = 0;
        for (int b = 0; b < nblk; ++b) { t0 = std::min(t0, h[b * 8]); t1 = std::max(t1, h[b * 8 + 4]); }
        double seg[4] = {0, 0, 0, 0};
        for (int b = 0; b < nblk; ++b)
            for (int k = 0; k < 4; ++k) seg[k] += (h[b * 8 + k + 1] - h[b * 8 + k]) * 0.01;
        printf("%s M=%d N=%d K=%d: %.1f us/launch (%.0f TF/s), %d blocks; stamped span %.1f us\n", sh.name, M, sh.N, sh.K, ms * 100,
               2.0 * M * sh.N * sh.K / (ms * 100) / 1e6, nblk, (t1 - t0) * 0.01);
        printf("   mean per block [us]: prologue %.2f  mainloop %.2f  epilogue-half0 %.2f  epilogue-half1 %.2f\n", seg[0] / nblk,
               seg[1] / nblk, seg[2] / nblk, seg[3] / nblk);
        // rounds: sort block start times
        std::vector<double> st(nblk), en(nblk);
        for (int b = 0; b < nblk; ++b) { st[b] = (h[b * 8] - t0) * 0.01; en[b] = (h[b * 8 + 4] - t0) * 0.01; }
        std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
        printf("   block starts  p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f | ends p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f\n", st[0],
               st[nblk / 4], st[nblk / 2], st[3 * nblk / 4], st[nblk - 1], en[0], en[nblk / 4], en[nblk / 2], en[3 * nblk / 4], en[nblk - 1]);
        // per-round detail for the first 3 blocks on one CU: find blocks sharing HW_ID cu/se/xcc with block 0
        const long long cu0 = h[5] & 0xff00, x0 = h[6];   // cu_id/sh/se bits
        {   // per-wave phase cycles of the k-loop (first 256 blocks): R work | wait at barrier 1 | M work | wait at barrier 2
            std::vector<long long> phv(256 * 8 * 4);
            hipMemcpy(phv.data(), tl + 4096 * 8, phv.size() * 8, hipMemcpyDeviceToHost);
            const int nb = nblk < 256 ? nblk : 256;
            double s4[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            for (int b = 0; b < nb; ++b)
                for (int w = 0; w < 8; ++w)
                    for (int k = 0; k < 4; ++k) s4[w >> 2][k] += (double)phv[((size_t)b * 8 + w) * 4 + k];
            const double steps = (double)nb * 4 * (sh.K / 32);
            for (int grp = 0; grp < 2; ++grp)
                printf("   group %d cycles per k-step: R %.0f | barrier-1 wait %.0f | M %.0f | barrier-2 wait %.0f\n", grp, s4[grp][0] / steps,
                       s4[grp][1] / steps, s4[grp][2] / steps, s4[grp][3] / steps);
        }
        printf("   blocks on block-0's CU:");
        for (int b = 0; b < nblk; ++b)
            if ((h[b * 8 + 5] & 0xff00) == cu0 && h[b * 8 + 6] == x0)
                printf(" [b%d %.1f|%.1f|%.1f|%.1f|%.1f]", b, (h[b * 8] - t0) * .01, (h[b * 8 + 1] - t0) * .01, (h[b * 8 + 2] - t0) * .01,
                       (h[b * 8 + 3] - t0) * .01, (h[b * 8 + 4] - t0) * .01);
        printf("\n");
        hipFree(w); hipFree(x); hipFree(ob); hipFree(of); hipFree(res); hipFree(bias);
    }
    return 0;
}
