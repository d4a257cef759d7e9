// What paces wstream_kernel (csrc/wstream.hip) on the gate|up shape of a 128-stream decode step (N = 28672 as SwiGLU-dual, K = 4096, M = 128)?  The same kernel
// with pieces removed: -DWS_X_NOMFMA=1 (pure streaming), -DWS_X_NOXREAD=1, -DWS_X_NOBAR=1.
//   for F in "" -DWS_X_NOMFMA=1 "-DWS_X_NOMFMA=1 -DWS_X_NOXREAD=1" "-DWS_X_NOMFMA=1 -DWS_X_NOXREAD=1 -DWS_X_NOBAR=1"; do hipcc --offload-arch=gfx950 -O3 -std=c++17 $F -Istreammind_amd/csrc -Iinclude tools/experiments/wstream_probe.hip -o /tmp/wsp && /tmp/wsp; done
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../streammind_amd/csrc/wstream.hip"
thread_local char g_sm_err[512];
int main() {
    struct { const char* name; int N, K, dual, S; } shapes[] = {{"gate|up dual", 28672, 4096, 1, 1}, {"down", 4096, 14336, 0, 8}, {"qkv", 6144, 4096, 0, 8}, {"qkv S=4", 6144, 4096, 0, 4}, {"o", 4096, 4096, 0, 8}};
    const int M = 128;
    for (auto& sh : shapes) {
        void *w, *x, *ob; float* ws;
        hipMalloc(&w, (size_t)sh.N * sh.K * 2); hipMalloc(&x, (size_t)M * sh.K * 2); hipMalloc(&ob, (size_t)M * sh.N * 4); hipMalloc(&ws, (size_t)8 * M * sh.N * 4);
        hipMemset(w, 0x11, (size_t)sh.N * sh.K * 2); hipMemset(x, 0x11, (size_t)M * sh.K * 2);
        LinArgs a; memset(&a, 0, sizeof(a));
        a.w = (const bf16x8*)w; a.N = sh.N; a.K = sh.K; a.KS = sh.K / 32; a.NRG = sh.N / 16; a.x = x; a.M = M; a.ldx = sh.K;
        a.act = sh.dual ? SM_ACT_SWIGLU_DUAL : 0; a.out_bf16 = sh.dual ? (bf16_t*)ob : nullptr; a.ldo_bf16 = sh.N / 2; a.out_f32 = sh.dual ? nullptr : (float*)ob; a.ldo = sh.N;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it = 0; it < 3; ++it) launch_wstream(a, 0, sh.S > 1 ? ws : nullptr, sh.S, a.KS / sh.S);
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) launch_wstream(a, 0, sh.S > 1 ? ws : nullptr, sh.S, a.KS / sh.S);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-14s N=%5d K=%5d S=%d: %6.1f us  %.2f TB/s of weights   [NOMFMA=%d NOXREAD=%d NOBAR=%d]\n", sh.name, sh.N, sh.K, sh.S, ms * 50, (double)sh.N * sh.K * 2 / (ms * 50e-6) / 1e12,
               WS_X_NOMFMA, WS_X_NOXREAD, WS_X_NOBAR);
        hipFree(w); hipFree(x); hipFree(ob); hipFree(ws);
    }
    return 0;
}
