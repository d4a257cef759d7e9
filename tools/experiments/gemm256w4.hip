// EXPERIMENT (not built into the library): round-1 state of the 4-wave 256x256 GEMM.  To try it again: move this file to
// streammind_amd/csrc/, add it to SOURCES in streammind_amd/build.py, move half_rows_epilogue from gemm256.hip to
// linear_common.h, declare launch_gemm256w4 there and call it from sm_linear's 256-tile branch.
// Measured (MI355X, M = 16156, bf16 out, TFLOP/s; qkv / out / fc1 / fc2 / 4096^2):
//   this kernel            924 / 902 / 983 / 1283 / 1268      (correct: same check error as gemm256.hip)
//   gemm256.hip (8 waves) 1055 / 1014 / 1066 / 1353 / 1359
//   hipBLASLt asm stream-K 1118 / 1112 / 1202 / 1455 / 1503   (F.linear; MT256x256x64, MIWT8_8 = this wave layout)
// What was learnt: (1) with the builtin MFMA hipcc keeps a third of the 64 accumulator tiles in VGPRs (120 v_accvgpr moves per
// k-step, 980 TFLOP/s); the inline-asm MFMA with a "+a" accumulator gives the loop as written (0 moves); (2) its first version
// was WRONG because the tail instantiations spilled and hipcc moved accumulators right behind asm MFMAs it cannot see (hazard):
// the tail now uses builtin MFMAs behind an s_nop pad; (3) 4 vs 5 ring stages, 2 / 4 / 8 / 16 MFMAs per scheduling group: no
// difference; (4) scalar DMA bases (+4 %); (5) what is left is the issue cost of 8 LDS-DMA pieces per wave per k-step (~60-100
// clk each) that no second wave on the SIMD hides any more -- the vendor kernel stages through registers (global_load ->
// ds_write) instead.  Next: that data path, or the DMA issued by a fifth, loader-only wave.
// 256(m) x 256(n) bf16 MFMA GEMM, FOUR waves per block: each wave owns 128(n) x 128(m) = 8 x 8 fragments (256 accumulator
// registers, pinned in the AGPR file; one wave per SIMD) -- the shape of hipBLASLt's assembly kernel for these sizes
// (MT256x256x64, MIWT8_8).  Same LDS image as gemm256.hip (ring of 32-KiB stages, one 32-deep k-step each, both operands by
// global_load_lds), but per k-step a wave reads 16 fragments for 64 MFMAs (4.0 MFMAs per ds_read_b128 against 2.67 in the
// 8-wave kernel) and there is ONE barrier per k-step.  With a single wave per SIMD nothing else hides latency: the fragments of
// k-step t+1 are read into a second register set while the MFMAs of k-step t run, in groups of {4 MFMAs, 1 fragment read,
// every other group 1 DMA piece} pinned by sched_barrier.
// The MFMAs of the main loop are inline asm with a "+a" accumulator: with the builtin, hipcc kept a third of the 64 tiles in
// VGPRs and shuttled them through a[0:3] (120 v_accvgpr moves per k-step).  The hazard recogniser does not see inside the asm,
// so compiler-generated moves of an accumulator right behind such an MFMA would read it too early: the loop has none (checked in
// the ISA), and the tail k-steps (plain builtin MFMAs) and the epilogue start behind an s_nop pad.
// Only the plain fast cases (no V^T side output, no row remap, full 256-column tiles); otherwise the launcher returns -1 and
// the caller keeps gemm256.hip.
#include <stdlib.h>
#include <type_traits>

#include "linear_common.h"

#define W4_BM 256
#define W4_BN 256
#define W4_STAGE 32768
#ifndef W4_GROUP
#define W4_GROUP 8           // MFMAs per scheduling group of the k-step (4, 8, 16, 32)
#endif

__device__ __forceinline__ void mfma_agpr(f32x4& c, const bf16x8& a, const bf16x8& b) {
    const u32x4 au = __builtin_bit_cast(u32x4, a), bu = __builtin_bit_cast(u32x4, b);
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(au), "v"(bu));
}

template <int ACT, int NST>      // NST: ring stages (4 = 128 KiB, 5 = 160 KiB)
__global__ __launch_bounds__(256, 1) void gemm256w4_kernel(LinArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // provably uniform: LDS destinations and DMA bases stay scalar
    const int i = lane & 15, g = lane >> 4;
    const int wn = wave >> 1, wm = wave & 1;

    const int nblk = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-banded, bijective tile order (block b runs on XCD b % 8)
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int KS = a.KS;
    constexpr int AHEAD = NST - 1;           // k-steps of DMA in flight ahead of the one being multiplied

    // ---- staging sources: 16 W pieces + 16 X pieces of 1 KiB per k-step, 4 + 4 per wave.  Every address is a wave-uniform
    // 64-bit base (scalar registers, advanced per k-step on the scalar unit) plus a 32-bit per-lane offset fixed for the whole
    // tile: no vector address arithmetic inside the k-loop (it cost ~5 VALU instructions per DMA piece between the MFMAs).
    const char* const wbase = (const char*)a.w + (size_t)(tile_n * 16 + wave * 4) * KS * 1024;        // N % 256 == 0: no clamp
    const uint32_t woff = lane * 16;
    const char* const xbase = (const char*)a.x + (size_t)tile_m * W4_BM * a.ldx * 2;
    uint32_t xoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 16 + (lane >> 2);
        int rl = row;
        if (tile_m * W4_BM + row >= a.M) rl = a.M - 1 - tile_m * W4_BM;
        const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
        xoff[j] = (uint32_t)(rl * a.ldx + chunk * 8) * 2;
    }
    auto stage_piece = [&](int ks, int slot, int q) {            // q = 0..3: W piece q, 4..7: X piece q - 4
        char* sb = smem + slot * W4_STAGE;
        if (q < 4) glds16(wbase + ((size_t)q * KS + ks) * 1024 + woff, sb + (wave * 4 + q) * 1024);
        else glds16(xbase + (size_t)ks * 64 + xoff[q - 4], sb + 16384 + (wave * 4 + q - 4) * 1024);
    };
    auto stage = [&](int ks, int slot) {
#pragma unroll
        for (int q = 0; q < 8; ++q) stage_piece(ks, slot, q);
    };
    auto read_one = [&](int slot, int q, bf16x8 (&wf)[8], bf16x8 (&xf)[8]) {      // q = 0..7: W fragment q, 8..15: X fragment q - 8
        const char* sw = smem + slot * W4_STAGE;
        if (q < 8) {
            wf[q] = *(const bf16x8*)(sw + (wn * 8 + q) * 1024 + lane * 16);
        } else {
            const int ml = wm * 128 + (q - 8) * 16 + i;
            xf[q - 8] = *(const bf16x8*)(sw + 16384 + ml * 64 + ((g ^ ((0 - (ml >> 2)) & 3)) * 16));
        }
    };
    auto read_frags = [&](int slot, bf16x8 (&wf)[8], bf16x8 (&xf)[8]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) read_one(slot, q, wf, xf);
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int nf = 0; nf < 8; ++nf)
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) acc[nf][mf] = f32x4{0, 0, 0, 0};

    // prologue: stages 0 .. AHEAD-1 in flight, stage 0 awaited, its fragments in set A
#pragma unroll
    for (int j = 0; j < AHEAD; ++j)
        if (j < KS) stage(j, j);
    if (KS >= AHEAD) { if (AHEAD == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 wfA[8], xfA[8], wfB[8], xfB[8];
    read_frags(0, wfA, xfA);

    // One steady-state k-step on the fragments (wc, xc) read during the previous one.  In flight at its top: the batches of
    // ks+1 .. ks+AHEAD-1 (8 loads each).  Stage ks+1 is awaited, barrier (everybody's has landed, and everybody is done reading
    // the slot of k-step ks-1, which stage ks+AHEAD overwrites), then 16 groups of {4 MFMAs, one fragment read of ks+1 into
    // (wn_, xn_), every other group one DMA piece of ks+AHEAD}.
    auto kstep = [&](int ks, int slot_next, int slot_dma, bf16x8 (&wc)[8], bf16x8 (&xc)[8], bf16x8 (&wn_)[8], bf16x8 (&xn_)[8]) {
        if (AHEAD == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int q = 0; q < 64 / W4_GROUP; ++q) {   // groups of {W4_GROUP MFMAs, their share of the 16 fragment reads and 8 DMA pieces}
#pragma unroll
            for (int e = 0; e < W4_GROUP; ++e) {
                const int t = q * W4_GROUP + e, mf = t >> 3, nf = t & 7;
                mfma_agpr(acc[nf][mf], wc[nf], xc[mf]);
            }
#pragma unroll
            for (int r = 0; r < W4_GROUP / 4; ++r) read_one(slot_next, q * (W4_GROUP / 4) + r, wn_, xn_);
            if (W4_GROUP >= 8) {
#pragma unroll
                for (int r = 0; r < W4_GROUP / 8; ++r) stage_piece(ks + AHEAD, slot_dma, q * (W4_GROUP / 8) + r);
            } else if ((q & 1) == 0) {
                stage_piece(ks + AHEAD, slot_dma, q >> 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int ks = 0, s0 = 0;                      // s0 = ring slot of k-step ks
    auto nxt = [](int s, int d) { s += d; return s >= NST ? s - NST : s; };
    for (; ks + AHEAD + 1 < KS; ks += 2) {   // two k-steps per trip: the register sets alternate without copies
        kstep(ks, nxt(s0, 1), nxt(s0, AHEAD), wfA, xfA, wfB, xfB);
        kstep(ks + 1, nxt(s0, 2), nxt(s0, AHEAD + 1 - NST + NST), wfB, xfB, wfA, xfA);
        s0 = nxt(s0, 2);
    }
    // ---- tail: the remaining k-steps (stages up to ks+AHEAD-1 are issued) with plain builtin MFMAs, fragments re-read
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the asm MFMAs' results settle before compiler code may touch them
    for (int j = ks; j < KS; ++j) {
        // stage j has landed; in flight may stay the younger batches already issued: j+1 .. min(j+AHEAD-1, KS-1)
        const int young = KS - 1 - j < AHEAD - 1 ? KS - 1 - j : AHEAD - 1;
        if (young >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (young == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (young == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j + AHEAD < KS) stage(j + AHEAD, nxt(s0, AHEAD));        // its slot held k-step j-1: everybody is past it
        bf16x8 wf[8], xf[8];
        read_frags(s0, wf, xf);
#pragma unroll
        for (int mf = 0; mf < 8; ++mf)
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
                acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xf[mf], acc[nf][mf], 0, 0, 0);
        s0 = nxt(s0, 1);
    }
    __syncthreads();       // every wave is done reading the ring before the epilogue reuses it

    // ---- epilogue
    if (a.out_bf16 && !a.out_f32 && !a.residual) {
        // bf16-only outputs: bias + activation in registers, the whole tile staged as bf16 ([256 m][256 n], 512-byte rows, 16-byte
        // chunk index XOR (m & 31)), 16 B per lane to memory
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) {
            const int nl = wn * 128 + nf * 16 + g * 4;
            f32x4 b4 = {0, 0, 0, 0};
            if (a.bias) b4 = *(const f32x4*)(a.bias + tile_n * W4_BN + nl);
#pragma unroll
            for (int mf = 0; mf < 8; ++mf) {
                const int ml = wm * 128 + mf * 16 + i;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[nf][mf][j] + b4[j];
                    if (ACT == SM_ACT_QUICK_GELU) t = t * sigmoidf_(1.702f * t);
                    o[j] = t;
                }
                *(u32x2*)(smem + ml * 512 + (((nl >> 3) ^ (ml & 31)) * 16) + (nl & 4) * 2) = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
            }
        }
        __syncthreads();
        bf16_t* const __restrict__ ob = a.out_bf16;
        const int chunk = tid & 31;
#pragma unroll
        for (int p0 = 0; p0 < 32; p0 += 4) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ml = (p0 + u) * 8 + (tid >> 5);
                v[u] = *(const u32x4*)(smem + ml * 512 + ((chunk ^ (ml & 31)) * 16));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = tile_m * W4_BM + (p0 + u) * 8 + (tid >> 5);
                if (m < a.M) store16_wt(ob + (size_t)m * a.ldo_bf16 + tile_n * W4_BN + chunk * 8, v[u]);
            }
        }
        return;
    }
    // fp32 (+ residual, + bf16 copy) outputs: two 128-row halves staged as fp32 ([128 m][256 n], 1 KiB rows), whole rows out
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        if (wm == half) {
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                for (int mf = 0; mf < 8; ++mf) {
                    const int ml = mf * 16 + i;
                    const int chunk = wn * 32 + nf * 4 + g;
                    *(f32x4*)(smem + ml * 1024 + ((chunk ^ (ml & 31)) * 16)) = acc[nf][mf];
                }
        }
        __syncthreads();
        half_rows_epilogue<ACT, 64, 4>(smem, a.bias, a.residual, a.ldr, a.out_f32, a.ldo, a.out_bf16, a.ldo_bf16,
                                       tile_m * W4_BM + half * 128, tile_n * W4_BN, a.M, tid);
    }
}

// returns SM_OK after launching, or -1 when the shape / epilogue is outside this kernel's fast cases (nothing launched)
int launch_gemm256w4(const LinArgs& a, int act, hipStream_t st) {
    if (a.vt || a.remap_in != 0 || (a.N % W4_BN) != 0 || (act != SM_ACT_NONE && act != SM_ACT_QUICK_GELU) || a.wscale) return -1;
    if ((a.ldo & 3) || (a.ldr & 3) || (a.ldo_bf16 & 7) || ((uintptr_t)a.out_bf16 & 15) || (!a.out_f32 && !a.out_bf16)) return -1;
    static int nst = -1;
    if (nst < 0) { const char* e = getenv("SM_GEMM_W4_STAGES"); nst = e ? atoi(e) : 4; }
    const int LDS = nst * W4_STAGE;
    const int tiles_m = cdiv(a.M, W4_BM), tiles_n = a.N / W4_BN;
    static bool attr_set = false;
    if (!attr_set) {
        SM_HIP(hipFuncSetAttribute((const void*)gemm256w4_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * W4_STAGE));
        SM_HIP(hipFuncSetAttribute((const void*)gemm256w4_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * W4_STAGE));
        SM_HIP(hipFuncSetAttribute((const void*)gemm256w4_kernel<0, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * W4_STAGE));
        SM_HIP(hipFuncSetAttribute((const void*)gemm256w4_kernel<1, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * W4_STAGE));
        attr_set = true;
    }
    const dim3 grid(tiles_m * tiles_n);
    if (nst == 5) {
        if (act == SM_ACT_NONE) gemm256w4_kernel<0, 5><<<grid, 256, LDS, st>>>(a, tiles_m, tiles_n);
        else gemm256w4_kernel<1, 5><<<grid, 256, LDS, st>>>(a, tiles_m, tiles_n);
    } else {
        if (act == SM_ACT_NONE) gemm256w4_kernel<0, 4><<<grid, 256, LDS, st>>>(a, tiles_m, tiles_n);
        else gemm256w4_kernel<1, 4><<<grid, 256, LDS, st>>>(a, tiles_m, tiles_n);
    }
    SM_LAUNCH_CHECK();
    return SM_OK;
}
