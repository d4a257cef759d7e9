// 256(m) x 256(n) x 64 bf16 MFMA GEMM for gfx950: the large-M workhorse (ViT batches, LLM prefill).
//
// Why a second tile size: the 128x128 kernel moves 32 KiB of operands through L2->LDS per 4.2 MFLOP (64 FLOP/B) and
// is bound by that traffic (~15 TB/s chip-wide measured) at ~0.5-0.75 PFLOP/s; 256x256 doubles the intensity
// (128 FLOP/B), so the same L2 bandwidth feeds twice the MFMA rate.
//
// Structure: 512 threads = 8 waves as 2(n) x 4(m); a wave owns 128(n) x 64(m) = 8 x 4 fragments (128 accumulator
// VGPRs).  One block per CU; LDS = ring of 4 slots x (W 16 KiB + X 16 KiB), one 32-deep k-step per slot, loads three
// k-steps ahead, one barrier per k-step (a 2-stage BK=64 version measured 1.7-2.4 us per 64-deep tile against
// 0.85 us of MFMA time: with one block per CU a single tile in flight cannot cover the L2 latency).  Both operands
// arrive by global_load_lds (16 B/lane): W chunks are already fragment-ordered (packed layout), X rows are
// XOR-swizzled on the SOURCE address.  Epilogue: the fp32 tile is staged through LDS in two 128-row halves and written
// as whole rows (1 KiB per wave-store), with the activation resolved at compile time and __restrict__ pointers so the
// passes are not serialised on store round trips (see linear.hip).
#include <stdlib.h>

#include "linear_common.h"

#define G2_BM 256
#define G2_BN 256
#define G2_SLOT 32768
#define G2_LDS (4 * G2_SLOT)

template <int ACT>
__device__ __forceinline__ void half_rows_epilogue(const char* __restrict__ smem, const float* __restrict__ bias,
                                                   const float* __restrict__ residual, int ldr, float* __restrict__ out_f32,
                                                   int ldo, bf16_t* __restrict__ out_bf16, int ldob, int m0, int n0, int M,
                                                   int tid) {
    const int chunk = tid & 63;
    const int n = n0 + chunk * 4;
    f32x4 b4 = {0, 0, 0, 0};
    if (bias) b4 = *(const f32x4*)(bias + n);
#pragma unroll
    for (int p0 = 0; p0 < 16; p0 += 4) {
        f32x4 v[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ml = (p0 + u) * 8 + (tid >> 6);
            v[u] = *(const f32x4*)(smem + ml * 1024 + ((chunk ^ (ml & 31)) * 16));
            r[u] = f32x4{0, 0, 0, 0};
            if (residual && m0 + ml < M) r[u] = *(const f32x4*)(residual + (size_t)(m0 + ml) * ldr + n);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + (p0 + u) * 8 + (tid >> 6);
            if (m >= M) continue;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[u][j] + b4[j];
                if (ACT == SM_ACT_QUICK_GELU) t = t * sigmoidf_(1.702f * t);
                o[j] = t + r[u][j];
            }
            if (out_f32) *(f32x4*)(out_f32 + (size_t)m * ldo + n) = o;
            if (out_bf16) *(u32x2*)(out_bf16 + (size_t)m * ldob + n) = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        }
    }
}

template <int ACT, int VAR = 0>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(LinArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int wn = wave >> 2, wm = wave & 3;

    const int nblk = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-banded, bijective tile order (block b runs on XCD b % 8)
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int KT = VAR == 1 ? 1 : a.KS;        // k-steps of 32

    // Staging: a ring of 4 LDS slots of 32 KiB, one 32-deep k-step each (W: 16 packed 1-KiB chunks; X: [256][32] bf16,
    // 64-byte rows, 16-byte chunk index XOR P[(row >> 2) & 3], P = {0,3,2,1}: conflict-free for the ds_read_b128 lane
    // groups).  Loads run THREE k-steps ahead of the MFMAs and there is ONE barrier per k-step:
    //   iteration i:  wait(own loads of slot i) -> barrier -> issue loads of step i+3 into the slot step i-1 just
    //                 vacated -> 12 ds_read_b128 + 32 MFMA on slot i.
    const char* wsrc[2];
    const char* xsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = wave * 2 + j;             // 0..15: W row-group / X piece of 16 rows
        int rgg = tile_n * 16 + c;
        if (rgg >= a.NRG) rgg = a.NRG - 1;
        wsrc[j] = (const char*)a.w + (size_t)rgg * a.KS * 1024 + lane * 16;
        const int row = c * 16 + (lane >> 2);
        int mg = tile_m * G2_BM + row;
        if (mg >= a.M) mg = a.M - 1;
        const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
        xsrc[j] = (const char*)a.x + ((size_t)mg * a.ldx + chunk * 8) * 2;
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int nf = 0; nf < 8; ++nf)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) acc[nf][mf] = f32x4{0, 0, 0, 0};

    auto stage = [&](int ks, int slot) {
        char* sb = smem + slot * G2_SLOT;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = wave * 2 + j;
            glds16(wsrc[j] + (size_t)ks * 1024, sb + c * 1024);
            glds16(xsrc[j] + (size_t)ks * 64, sb + 16384 + c * 1024);
        }
    };

    stage(0, 0);
    if (KT > 1) stage(1, 1);
    if (KT > 2) stage(2, 2);
    for (int ks = 0; ks < KT; ++ks) {
        const int rem = KT - 1 - ks;            // k-steps whose loads may still be in flight behind this one
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ks + 3 < KT) stage(ks + 3, (ks + 3) & 3);
        const char* sw = smem + (ks & 3) * G2_SLOT;
        const char* sx = sw + 16384;
        bf16x8 xf[4];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int ml = wm * 64 + mf * 16 + i;
            xf[mf] = *(const bf16x8*)(sx + ml * 64 + ((g ^ ((0 - (ml >> 2)) & 3)) * 16));
        }
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) {
            const bf16x8 wf = *(const bf16x8*)(sw + (wn * 8 + nf) * 1024 + lane * 16);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
                acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[mf], acc[nf][mf], 0, 0, 0);
        }
    }
    __syncthreads();       // every wave is done reading the ring before the epilogue reuses it

    if (VAR == 2) {
        float t = 0.f;
#pragma unroll
        for (int nf = 0; nf < 8; ++nf)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) t += acc[nf][mf][0] + acc[nf][mf][1] + acc[nf][mf][2] + acc[nf][mf][3];
        if (t == 123.456f && a.out_f32) a.out_f32[0] = t;
        return;
    }
    // ---- epilogue in two 128-row halves: waves wm = 2h, 2h+1 stage their accumulators ([128 m][256 n] fp32 = 128 KiB,
    // 16-byte chunk index XOR (m & 31)), then all 8 waves write whole rows.
    const bool vt_tile = a.vt && tile_n * G2_BN >= a.vt_n0;
    const bool fast = ACT >= 0 && a.remap_in == 0 && (tile_n + 1) * G2_BN <= a.N && (a.ldo & 3) == 0 && (a.ldo_bf16 & 3) == 0 &&
                      (a.ldr & 3) == 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        if ((wm >> 1) == half) {
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {
                    const int ml = (wm & 1) * 64 + mf * 16 + i;
                    const int chunk = wn * 32 + nf * 4 + g;
                    *(f32x4*)(smem + ml * 1024 + ((chunk ^ (ml & 31)) * 16)) = acc[nf][mf];
                }
        }
        __syncthreads();
        const int m0 = tile_m * G2_BM + half * 128;
        if (vt_tile) {
            const int nh = (a.N - a.vt_n0) / a.vt_dh;
            for (int pass = 0; pass < 32; ++pass) {
                const int nl = pass * 8 + wave;
                const int n = tile_n * G2_BN + nl;
                if (n >= a.N) continue;
                const int c = n - a.vt_n0;
                const int h = c / a.vt_dh, d = c - h * a.vt_dh;
                const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int ml = lane + 64 * h2;
                    const int m = m0 + ml;
                    if (m < a.M) {
                        float v = *(const float*)(smem + ml * 1024 + (((nl >> 2) ^ (ml & 31)) * 16) + (nl & 3) * 4) + bv;
                        const int b = m / a.vt_S, sidx = m - b * a.vt_S;
                        a.vt[((size_t)(b * nh + h) * a.vt_dh + d) * a.vt_ld + sidx] = (bf16_t)f2bf(apply_act_rt(v, a.act));
                    }
                }
            }
        } else if (fast) {
            half_rows_epilogue<ACT>(smem, a.bias, a.residual, a.ldr, a.out_f32, a.ldo, a.out_bf16, a.ldo_bf16, m0,
                                    tile_n * G2_BN, a.M, tid);
        } else {
            for (int pass = 0; pass < 16; ++pass) {
                const int ml = pass * 8 + (tid >> 6);
                const int chunk = tid & 63;
                f32x4 v = *(const f32x4*)(smem + ml * 1024 + ((chunk ^ (ml & 31)) * 16));
                store4(a, m0 + ml, tile_n * G2_BN + chunk * 4, v, nullptr);
            }
        }
    }
}

int launch_gemm256(const LinArgs& a, int act, hipStream_t st) {
    const int tiles_m = cdiv(a.M, G2_BM), tiles_n = cdiv(a.N, G2_BN);
    static bool attr_set = false;
    if (!attr_set) {
        SM_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS));
        SM_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS));
        SM_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS));
        attr_set = true;
    }
    const dim3 grid(tiles_m * tiles_n);
    static int var = -1;
    if (var < 0) { const char* e = getenv("SM_G256_VAR"); var = e ? atoi(e) : 0; }
    if (var == 1) { hipFuncSetAttribute((const void*)gemm256_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS); gemm256_kernel<0, 1><<<grid, 512, G2_LDS, st>>>(a, tiles_m, tiles_n); return SM_OK; }
    if (var == 2) { hipFuncSetAttribute((const void*)gemm256_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS); gemm256_kernel<0, 2><<<grid, 512, G2_LDS, st>>>(a, tiles_m, tiles_n); return SM_OK; }
    if (act == SM_ACT_NONE) gemm256_kernel<0><<<grid, 512, G2_LDS, st>>>(a, tiles_m, tiles_n);
    else if (act == SM_ACT_QUICK_GELU) gemm256_kernel<1><<<grid, 512, G2_LDS, st>>>(a, tiles_m, tiles_n);
    else gemm256_kernel<-1><<<grid, 512, G2_LDS, st>>>(a, tiles_m, tiles_n);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
