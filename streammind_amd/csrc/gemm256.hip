// 256(m) x 256(n) bf16 MFMA GEMM for gfx950: the large-M workhorse (ViT batches, LLM prefill).
//
// Why a second tile size: the 128x128 kernel moves 32 KiB of operands through L2->LDS per 4.2 MFLOP (64 FLOP/B) and
// is bound by that traffic (~15 TB/s chip-wide measured) at ~0.5-0.75 PFLOP/s; 256x256 doubles the intensity
// (128 FLOP/B), so the same L2 bandwidth feeds twice the MFMA rate.
//
// Structure: 512 threads = 8 waves as 2(n) x 4(m); a wave owns 128(n) x 64(m) = 8 x 4 fragments (128 accumulator
// VGPRs).  One block per CU; LDS = ring of 4 stages x (W 16 KiB + X 16 KiB), one 32-deep k-step per stage, loads three
// k-steps ahead.  Both operands arrive by global_load_lds (16 B/lane): W chunks are already fragment-ordered (packed
// layout), X rows are XOR-swizzled on the SOURCE address.  The two wave columns run one barrier apart so that one of the
// two waves of every SIMD multiplies while the other reads fragments and issues the DMA (see the main loop).
// Measured per 64-deep k-tile per CU (MI355X, M=16156): lockstep 2-stage BK=64 loop 1.63-1.80 us, this loop 1.45-1.55 us
// (0.85 us = MFMA issue at 2.4 GHz); 4096^3: 1146 -> 1283 TFLOP/s.
// Epilogues: bf16-only outputs are converted in registers, staged whole-tile as bf16 and written 16 B per lane; fp32 /
// residual outputs are staged as fp32 in two 128-row halves and written as whole rows.  All stores are write-through
// (sc1), the activation is resolved at compile time and the pointers are __restrict__ so the passes are not serialised
// on store round trips (see linear.hip).  tools/gemm_timeline.hip prints the per-block phase times.
// Round 3: gemm256p_kernel (below) is the PERSISTENT form of the same 256 x 256 loop for bf16-only outputs with several tiles per
// CU -- one block per CU walks its tiles and keeps the LDS-DMA pipeline running across the seam between two tiles (in situ: fc1
// 138 -> 125 us, QKV 96 -> 87 us on the same box; bit-identical results).
#include <stdlib.h>
#include <type_traits>

#include "linear_common.h"

#define G2_BM 256
#define G2_BN 256
// 256 x 256 tile: v_mfma_f32_16x16x32 (0) or v_mfma_f32_32x32x16 (1).  Round 3, same box, M = 16156 (tools/ab_gemm.sh): the 32x32x16
// loop is 4-7 % SLOWER on every shape (qkv 97.8 -> 102.3 us, fc1 128.8 -> 138, 4096^2 1413 -> 1318 TFLOP/s) although it issues half
// the MFMA instructions over the same 12 conflict-free fragment reads: the loop is not bound by MFMA issue slots.  Kept for A/B.
#ifndef G2_MFMA32
#define G2_MFMA32 0
#endif
// bf16-only outputs of the 256 x 256 tile written straight from the accumulators (1) or staged through LDS (0)
#ifndef G2_DIRECT_EPI
#define G2_DIRECT_EPI 0
#endif

// tools/gemm_timeline.hip compiles this file with SM_GEMM_TIMELINE to stamp each block's phases (100 MHz wall clock)
#ifdef SM_GEMM_TIMELINE
__device__ long long* g_gemm_timeline;
#define TL(k) do { if (threadIdx.x == 0) g_gemm_timeline[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TL(k) do {} while (0)
#endif

template <int ACT, int CPR, int RPP, bool F16>     // CPR: 16-byte chunks per staged row; RPP: rows covered by one pass of the block
__device__ __forceinline__ void half_rows_epilogue(const char* __restrict__ smem, const float* __restrict__ bias,
                                                   const float* __restrict__ residual, int ldr, float* __restrict__ out_f32,
                                                   int ldo, bf16_t* __restrict__ out_bf16, int ldob, int m0, int n0, int M,
                                                   int tid) {
    const int chunk = tid % CPR;
    const int n = n0 + chunk * 4;
    f32x4 b4 = {0, 0, 0, 0};
    if (bias) b4 = *(const f32x4*)(bias + n);
#pragma unroll
    for (int p0 = 0; p0 < 128 / RPP; p0 += 4) {
        f32x4 v[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ml = (p0 + u) * RPP + tid / CPR;
            v[u] = *(const f32x4*)(smem + ml * (CPR * 16) + ((chunk ^ (ml & 31)) * 16));
            r[u] = f32x4{0, 0, 0, 0};
            if (residual && m0 + ml < M) r[u] = *(const f32x4*)(residual + (size_t)(m0 + ml) * ldr + n);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + (p0 + u) * RPP + tid / CPR;
            if (m >= M) continue;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[u][j] + b4[j];
                if (ACT == SM_ACT_QUICK_GELU) t = t * sigmoidf_(1.702f * t);
                o[j] = t + r[u][j];
            }
            if (out_f32) store16_wt(out_f32 + (size_t)m * ldo + n, __builtin_bit_cast(u32x4, o));
            if (out_bf16) *(u32x2*)(out_bf16 + (size_t)m * ldob + n) = u32x2{pack16<F16>(o[0], o[1]), pack16<F16>(o[2], o[3])};
        }
    }
}

// sum over the 32 lanes of a half wave on the VALU: two quad permutes, the two row mirrors, one v_permlane16_swap (fixed order: deterministic)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_wave_sum(float v) {
    v = dpp_add<0xB1>(v);          // quad_perm [1, 0, 3, 2]
    v = dpp_add<0x4E>(v);          // quad_perm [2, 3, 0, 1]
    v = dpp_add<0x141>(v);         // row_half_mirror: quads 0 <-> 1, 2 <-> 3
    v = dpp_add<0x140>(v);         // row_mirror: the two halves of a 16-lane row
    return xor16_sum(v);           // rows 0 <-> 1 (2 <-> 3)
}

// The fp32 + residual epilogue of the 256 x 256 tile as the PRODUCER of a folded LayerNorm (sm_linear_t.fold_stats_out; 512 threads, 128 staged rows
// of 256 fp32 columns).  32 lanes own one row, 16 rows per pass, 8 passes per half tile.  Lane t holds the row's 16-byte chunks t and t + 32, so every
// load / store INSTRUCTION covers contiguous memory (512 B of one row per half wave): the first form of this epilogue gave a lane 8 consecutive columns
// -- two 16-byte fp32 stores per lane at a 32-byte stride, i.e. half-filled 64-byte segments twice over -- and cost +12 us per launch, what write-through
// stores do with partial segments.  The 16-bit copy needs 8 consecutive columns per 16-byte store: lane pairs (t, t ^ 1) swap one packed chunk on the VALU
// (quad permute), after which even lanes hold columns 8t' .. 8t' + 7 of the row's first half and odd lanes of its second half -- one store instruction, 512
// contiguous bytes per row.  The row's sum / sum of squares over the tile's 256 columns is reduced on the VALU and written by lane 0.
template <bool F16>
__device__ __forceinline__ void half_rows_epilogue_fold(const char* __restrict__ smem, const float* __restrict__ bias, const float* __restrict__ residual, int ldr,
                                                        float* __restrict__ out_f32, int ldo, bf16_t* __restrict__ out_ht, int ldh, const float* __restrict__ og,
                                                        float* __restrict__ ostats, int ntiles, int tile_n, int m0, int M, int tid) {
    const int t = tid & 31, rsub = tid >> 5;
    const int n0 = tile_n * 256 + t * 4, n1 = n0 + 128;             // the lane's two chunks: columns n0 .. n0 + 3 and n1 .. n1 + 3
    f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    if (bias) { b0 = *(const f32x4*)(bias + n0); b1 = *(const f32x4*)(bias + n1); }
    const f32x4 g0 = *(const f32x4*)(og + n0), g1 = *(const f32x4*)(og + n1);
    const bool odd = t & 1;
    // after the pair swap: even lane t -> columns 4t .. 4t + 7 (chunks t, t + 1); odd lane t -> columns 128 + 4(t - 1) .. + 7 (chunks t + 31, t + 32)
    const int nh = tile_n * 256 + (odd ? 128 + (t - 1) * 4 : t * 4);
#pragma unroll
    for (int p0 = 0; p0 < 8; p0 += 2) {         // two rows (32 bytes of fp32 + 32 of residual per lane and row) in flight: the other half tile's accumulators are still live
        f32x4 v0[2], v1[2], r0[2], r1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ml = (p0 + u) * 16 + rsub;
            v0[u] = *(const f32x4*)(smem + ml * 1024 + ((t ^ (ml & 31)) * 16));
            v1[u] = *(const f32x4*)(smem + ml * 1024 + (((t + 32) ^ (ml & 31)) * 16));
            r0[u] = r1[u] = f32x4{0, 0, 0, 0};
            if (residual && m0 + ml < M) {
                r0[u] = *(const f32x4*)(residual + (size_t)(m0 + ml) * ldr + n0);
                r1[u] = *(const f32x4*)(residual + (size_t)(m0 + ml) * ldr + n1);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int m = m0 + (p0 + u) * 16 + rsub;
            f32x4 o0, o1;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o0[j] = (v0[u][j] + b0[j]) + r0[u][j]; o1[j] = (v1[u][j] + b1[j]) + r1[u][j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1 += o0[j]; s2 = __builtin_fmaf(o0[j], o0[j], s2); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1 += o1[j]; s2 = __builtin_fmaf(o1[j], o1[j], s2); }
            s1 = half_wave_sum(s1);
            s2 = half_wave_sum(s2);
            // 16-bit(o * gamma): chunk t in (lo0, lo1), chunk t + 32 in (hi0, hi1); the even lane hands its second chunk to the odd lane and takes the odd lane's first
            const uint32_t lo0 = pack16<F16>(o0[0] * g0[0], o0[1] * g0[1]), lo1 = pack16<F16>(o0[2] * g0[2], o0[3] * g0[3]);
            const uint32_t hi0 = pack16<F16>(o1[0] * g1[0], o1[1] * g1[1]), hi1 = pack16<F16>(o1[2] * g1[2], o1[3] * g1[3]);
            const uint32_t x0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? lo0 : hi0), 0xB1, 0xf, 0xf, true);       // quad_perm [1, 0, 3, 2]: the pair partner's word
            const uint32_t x1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? lo1 : hi1), 0xB1, 0xf, 0xf, true);
            const u32x4 hv = odd ? u32x4{x0, x1, hi0, hi1} : u32x4{lo0, lo1, x0, x1};
            if (m < M) {
                store16_wt(out_f32 + (size_t)m * ldo + n0, __builtin_bit_cast(u32x4, o0));
                store16_wt(out_f32 + (size_t)m * ldo + n1, __builtin_bit_cast(u32x4, o1));
                store16_wt(out_ht + (size_t)m * ldh + nh, hv);
                if (t == 0) *(f32x2*)(ostats + ((size_t)m * ntiles + tile_n) * 2) = f32x2{s1, s2};
            }
        }
    }
}

template <int ACT, int WN, bool F16, bool PFOLD = false>   // WN = wave columns along n: tile is 256(m) x 128*WN(n), 4*WN waves; F16: fp16 operands / outputs; PFOLD: producer of a folded LayerNorm (its own instantiation: inlined beside the plain epilogue it cost that kernel 192 bytes of scratch)
__global__ __launch_bounds__(256 * WN, 2) void gemm256_kernel(LinArgs a, int tiles_m, int tiles_n, int cb) {
    constexpr int BN = 128 * WN;
    constexpr int NW = 4 * WN;                       // waves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably uniform: LDS-DMA destinations (M0) and source bases stay scalar
    const int i = lane & 15, g = lane >> 4;
    const int wn = wave >> 2, wm = wave & 3;

    const int nblk = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-banded, bijective tile order (block b runs on XCD b % 8)
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    if (cb > 0) {
        // blocked walk inside the XCD's band (rows_x row tiles x tiles_n column tiles): column groups of `cb` tiles, all rows of
        // the band inside a group -- the ~32 tiles an XCD runs at once then share cb/tiles_n of W and the band's X rows, so W
        // crosses the fabric once per XCD instead of once per round (profiles/r02_gemm_traffic.json)
        const int q = nblk >> 3, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int rows_x = q / tiles_n;
        const int cg = idx / (rows_x * cb), rem = idx - cg * rows_x * cb;
        const int r = rem / cb;
        tile_m = xcd * rows_x + r;
        tile_n = cg * cb + (rem - r * cb);
    }

    TL(0);
#ifdef SM_GEMM_TIMELINE
    if (threadIdx.x == 0) {
        g_gemm_timeline[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
        g_gemm_timeline[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
    }
#endif
    // Accumulators of a wave's 128(n) x 64(m) sub-tile.  16x16x32 layout: acc[nf][mf], lane (g, i) holds n = nf*16 + g*4 + 0..3 of
    // row m = mf*16 + i.  32x32x16 layout (256 x 256 tile): acc32[nb][mb], register q*4 + j of lane l holds n = nb*32 + q*8 +
    // (l / 32)*4 + j of row m = mb*32 + l % 32.  Either way a lane owns NP "pieces" of 4 consecutive n in each of NMB row blocks;
    // the epilogues below only see pieces.
    constexpr bool M32 = (WN == 2) && G2_MFMA32;
    constexpr int NP = M32 ? 16 : 8, NMB = M32 ? 2 : 4;      // pieces per row block, row blocks
    constexpr int PS = M32 ? 8 : 16, MBS = M32 ? 32 : 16;    // n stride between pieces, rows per row block
    const int lane_m = M32 ? (lane & 31) : i;
    const int lane_n = M32 ? (lane >> 5) * 4 : g * 4;
    f32x4 acc[8][4];
    f32x16 acc32[4][2];
    if constexpr (M32) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc32[nb][mb][j] = 0.f;
    } else {
#pragma unroll
        for (int nf = 0; nf < 8; ++nf)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) acc[nf][mf] = f32x4{0, 0, 0, 0};
    }
    auto piece = [&](int p, int mb) -> f32x4 {
        if constexpr (M32) {
            const int q = (p & 3) * 4;
            return f32x4{acc32[p >> 2][mb][q], acc32[p >> 2][mb][q + 1], acc32[p >> 2][mb][q + 2], acc32[p >> 2][mb][q + 3]};
        } else {
            return acc[p][mb];
        }
    };

    if constexpr (WN == 2) {
        // ---- 256 x 256: ring of four 32-KiB stages, one 32-deep k-step each (W: 16 packed 1-KiB fragment chunks; X: [256][32]
        // bf16, 64-byte rows, 16-byte chunk index XOR P[(row >> 2) & 3], P = {0,3,2,1}, applied on the SOURCE address of the
        // LDS-DMA and again on the read).  Loads run three k-steps ahead (4 global_load_lds per wave per step).
        //
        // Two wave groups (wn = 0 / 1: one wave of each per SIMD) run the SAME code one barrier apart, and every k-step is two
        // barrier-delimited intervals  R | M  (R = 12 fragment reads + the 4 DMA issues of step t+3, M = 32 MFMAs): while one
        // group multiplies, the other reads and issues loads, so a SIMD's matrix pipe and its LDS / address path work at
        // the same time instead of alternately (lockstep BK=64 version: ~3600 clk per 64-deep tile against 2048 clk of MFMA
        // issue; a staggered BK=64 version still put all 8 DMA issues, ~100 clk each, into one of its four intervals).
        // Hazards (intervals numbered globally; group 0 does R(t) in 2t and M(t) in 2t+1, group 1 one interval later):
        //   WAR  step t+3 is DMA'd into the stage of step t-1 during R(t) (interval >= 2t); that stage's last reader, group 1's
        //        R(t-1) in interval 2t-1, retired its reads (lgkmcnt(0)) before the barrier that opens interval 2t.
        //   RAW  before the barrier that closes its R(t) a wave has waited for its own loads of step t+1 (counted vmcnt: the
        //        batches of t+2 and t+3 may stay in flight); group 0 first reads step t+1 in interval 2t+2, which opens with
        //        the barrier group 1 reaches from its R(t) with that wait behind it.
        constexpr int STAGE = 32768;
        // split-K (gridDim.y slabs; round 5: the N = 4096 products of an LLM prefill chunk leave 64..128 tiles for 256 CUs): slab y multiplies
        // k-steps [y KS / S, (y + 1) KS / S) and leaves its raw accumulators in out_f32 + y M ldo (the launcher points out_f32 at the slabs, no
        // bias / residual); a.KS stays the row-group stride of the packed weights
        const int KS = a.KS / (int)gridDim.y;
        const int k0 = (int)blockIdx.y * KS;
        // Staging sources as a wave-uniform 64-bit base (scalar registers, advanced on the scalar unit) plus a 32-bit per-lane
        // offset that is fixed for the whole tile: no vector arithmetic per DMA piece.  VALU instructions of the wave that is
        // in its R interval are not free for the OTHER wave of the SIMD: on gfx950 they do not overlap with its MFMAs
        // (tools/experiments/mfma_valu_coexec2.hip), so the ~15 address / readfirstlane instructions per k-step the per-lane
        // pointers cost came straight out of the M interval (32 MFMAs measured at ~20 clk each instead of 16-17).
        const char* wbase[2];
        const uint32_t woff = lane * 16;
        const char* const xbase = (const char*)a.x + (size_t)tile_m * G2_BM * a.ldx * 2 + (size_t)k0 * 64;
        uint32_t xoff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int rgg = tile_n * 16 + wave * 2 + j;
            // SwiGLU-dual: the tile's 16 row groups alternate gate / up groups of the SAME 128 output columns (w = [gate rows | up rows]),
            // so fragments 2f and 2f + 1 of a wave hold gate and up of the same (row, column) in the same lane and register
            if constexpr (ACT == SM_ACT_SWIGLU_DUAL) rgg = ((wave * 2 + j) & 1) * (a.NRG >> 1) + tile_n * 8 + ((wave * 2 + j) >> 1);
            if (rgg >= a.NRG) rgg = a.NRG - 1;
            wbase[j] = (const char*)a.w + ((size_t)rgg * a.KS + k0) * 1024;
            const int row = (wave * 2 + j) * 16 + (lane >> 2);
            int rl = row;
            if (tile_m * G2_BM + row >= a.M) rl = a.M - 1 - tile_m * G2_BM;
            const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
            xoff[j] = (uint32_t)(rl * a.ldx + chunk * 8) * 2;
        }
        // global_load_lds in its scalar-base form (s[base] + v offset), written out: the builtin takes a per-lane 64-bit pointer and
        // hipcc rebuilds one with two v_lshl_add_u64 per piece even from a uniform base.  Waits are counted by hand below anyway.
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
        auto dma = [&](const char* sbase, uint32_t voff, uint32_t dst) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(dst) : "memory");
        };
        auto stage = [&](int ks, int slot) {
            const uint32_t sb = lds0 + slot * STAGE + wave * 2048;
#pragma unroll
            for (int j = 0; j < 2; ++j) dma(wbase[j] + (size_t)ks * 1024, woff, sb + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) dma(xbase + (size_t)ks * 64, xoff[j], sb + 16384 + j * 1024);
        };
        stage(0, 0);
        if (KS > 1) stage(1, 1);
        if (KS > 2) stage(2, 2);
        if (KS > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (KS > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        TL(1);
        if (wn == 1) __builtin_amdgcn_s_barrier();
        // one k-step; TAIL = 0: steady state (issue step ks+3, leave the batches of ks+2 and ks+3 in flight), 1 / 2: the last
        // steps (nothing left to issue).  A compile-time switch: runtime branches around the waits cost ~8 % in this loop.
#ifdef SM_GEMM_TIMELINE
        long long ph[4] = {0, 0, 0, 0}, tp = clock64();
#define PH(k) do { const long long t_ = clock64(); ph[k] += t_ - tp; tp = t_; } while (0)
#else
#define PH(k) do {} while (0)
#endif
        auto kstep = [&](int ks, auto tail, int slot) {      // slot = ks & 3
            constexpr int TAIL = decltype(tail)::value;
            const char* sw = smem + slot * STAGE;
            const char* sx = sw + 16384;
            // 16x16x32: 4 X + 8 W fragments of the whole 32-deep step.  32x32x16: per 16-deep half h, 2 X fragments (lane l: row
            // l % 32 of the row block, 16-byte chunk 2h + l / 32 of the 64-byte row) and 4 W fragments (rows l % 32 of a 32-row pair
            // of packed chunks: chunk (l % 32) / 16, its lane slot (2h + l / 32) * 16 + l % 16) -- the SAME LDS image, 12 reads
            // either way, all conflict-free (each ds_read_b128 lane group meets 16 distinct 16-byte slots mod 256 B).
            bf16x8 xf[4], wf[8];
            if constexpr (M32) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const int ml = wm * 64 + mb * 32 + (lane & 31);
                        xf[h * 2 + mb] = *(const bf16x8*)(sx + ml * 64 + (((2 * h + (lane >> 5)) ^ ((0 - (ml >> 2)) & 3)) * 16));
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        wf[h * 4 + nb] = *(const bf16x8*)(sw + (wn * 8 + nb * 2 + ((lane & 31) >> 4)) * 1024 +
                                                          ((2 * h + (lane >> 5)) * 16 + (lane & 15)) * 16);
            } else {
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {
                    const int ml = wm * 64 + mf * 16 + i;
                    xf[mf] = *(const bf16x8*)(sx + ml * 64 + ((g ^ ((0 - (ml >> 2)) & 3)) * 16));
                }
#pragma unroll
                for (int nf = 0; nf < 8; ++nf) wf[nf] = *(const bf16x8*)(sw + (wn * 8 + nf) * 1024 + lane * 16);
            }
            if (TAIL == 0) {
                stage(ks + 3, (slot + 3) & 3);
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            } else if (TAIL == 1) {
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            PH(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            PH(1);
            // no s_setprio around the MFMA cluster: measured 3-5 % slower with it in this two-group schedule (tools/gemm_timeline)
            if constexpr (M32) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb)
                            acc32[nb][mb] = mfma32<F16>(wf[h * 4 + nb], xf[h * 2 + mb], acc32[nb][mb]);
            } else {
#pragma unroll
                for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf)
                        acc[nf][mf] = mfma16<F16>(wf[nf], xf[mf], acc[nf][mf]);
            }
            PH(2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            PH(3);
        };
        int ks = 0;
        // four k-steps per trip with compile-time ring slots: the fragment reads take immediate offsets (no per-step address adds)
        for (; ks + 6 < KS; ks += 4) {
            kstep(ks, std::integral_constant<int, 0>{}, 0);
            kstep(ks + 1, std::integral_constant<int, 0>{}, 1);
            kstep(ks + 2, std::integral_constant<int, 0>{}, 2);
            kstep(ks + 3, std::integral_constant<int, 0>{}, 3);
        }
        for (; ks + 3 < KS; ++ks) kstep(ks, std::integral_constant<int, 0>{}, ks & 3);
        if (ks + 2 < KS) { kstep(ks, std::integral_constant<int, 1>{}, ks & 3); ++ks; }
        for (; ks < KS; ++ks) kstep(ks, std::integral_constant<int, 2>{}, ks & 3);
        if (wn == 0) __builtin_amdgcn_s_barrier();
#ifdef SM_GEMM_TIMELINE
        if (lane == 0 && blockIdx.x < 256) {
            for (int k = 0; k < 4; ++k) g_gemm_timeline[(size_t)4096 * 8 + ((size_t)blockIdx.x * 8 + wave) * 4 + k] = ph[k];
        }
#endif
    } else {
        // ---- 256 x 128 (two blocks per CU): ring of 3 LDS slots, one 32-deep k-step each (W: packed 1-KiB chunks;
        // X: [256][32] bf16, 64-byte rows, chunk index XOR P[(row >> 2) & 3], P = {0,3,2,1}); loads run two k-steps
        // ahead, one barrier per k-step.
        constexpr int WBYTES = BN * 64;
        constexpr int SLOT = WBYTES + 16384;
        constexpr int NSLOT = 3;
        constexpr int XL = 16 / NW;
        const int KT = a.KS;
        const char* wsrc[2];
        const char* xsrc[XL];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int rgg = tile_n * (BN / 16) + wave * 2 + j;
            if (rgg >= a.NRG) rgg = a.NRG - 1;
            wsrc[j] = (const char*)a.w + (size_t)rgg * a.KS * 1024 + lane * 16;
        }
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int row = (wave * XL + j) * 16 + (lane >> 2);
            int mg = tile_m * G2_BM + row;
            if (mg >= a.M) mg = a.M - 1;
            const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
            xsrc[j] = (const char*)a.x + ((size_t)mg * a.ldx + chunk * 8) * 2;
        }
        auto stage = [&](int ks, int slot) {
            char* sb = smem + slot * SLOT;
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(wsrc[j] + (size_t)ks * 1024, sb + (wave * 2 + j) * 1024);
#pragma unroll
            for (int j = 0; j < XL; ++j) glds16(xsrc[j] + (size_t)ks * 64, sb + WBYTES + (wave * XL + j) * 1024);
        };
#pragma unroll
        for (int pre = 0; pre < NSLOT - 1; ++pre)
            if (pre < KT) stage(pre, pre);
        int slot = 0;
        for (int ks = 0; ks < KT; ++ks) {
            if (KT - 1 - ks >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (ks + NSLOT - 1 < KT) stage(ks + NSLOT - 1, slot == 0 ? NSLOT - 1 : slot - 1);
            const char* sw = smem + slot * SLOT;
            const char* sx = sw + WBYTES;
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
                    acc[nf][mf] = mfma16<F16>(wf, xf[mf], acc[nf][mf]);
            }
            slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
    }
    __syncthreads();       // every wave is done reading the staging buffers before the epilogue reuses them
    TL(2);
    if constexpr (WN == 2) a.out_f32 += (size_t)blockIdx.y * a.M * a.ldo;       // split-K slab of this block (gridDim.y == 1: nothing)

    // ---- epilogue in two 128-row halves: waves wm = 2h, 2h+1 stage their accumulators ([128 m][BN n] fp32, 16-byte
    // chunk index XOR (m & 31)), then all waves write whole rows.
    constexpr int ROWB = BN * 4;                     // bytes per staged row
    constexpr int CPR = BN / 4;                      // 16-byte chunks per row
    constexpr int RPP = (256 * WN) / CPR;            // rows per pass (= 8)
    const bool vt_tile = a.vt && tile_n * BN >= a.vt_n0;
    const bool fast = ACT >= 0 && a.remap_in == 0 && (tile_n + 1) * BN <= a.N && (a.ldo & 3) == 0 && (a.ldo_bf16 & 3) == 0 &&
                      (a.ldr & 3) == 0;
    if constexpr (WN == 2 && ACT == SM_ACT_SWIGLU_DUAL) {
        // out_bf16[m][tile_n * 128 + c] = 16-bit(silu(gate + b_g) * (up + b_u)): in-lane (fragments 2f / 2f + 1), the 256 x 128 tile staged
        // as 16-bit ([256 m][128 n], 256-byte rows, 16-byte chunk index XOR (m & 15)) and written 16 B per lane.  The launcher guarantees
        // full column tiles, 16-byte aligned rows and no other output.
        static_assert(!M32, "SwiGLU-dual epilogue: 16x16x32 accumulators");
        const int F = a.N >> 1;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int nl = wn * 64 + f * 16 + g * 4;
            f32x4 bg = {0, 0, 0, 0}, bu = {0, 0, 0, 0};
            if (a.bias) { bg = *(const f32x4*)(a.bias + tile_n * 128 + nl); bu = *(const f32x4*)(a.bias + F + tile_n * 128 + nl); }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int ml = wm * 64 + mb * 16 + i;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = siluf_(acc[2 * f][mb][j] + bg[j]) * (acc[2 * f + 1][mb][j] + bu[j]);
                *(u32x2*)(smem + ml * 256 + (((nl >> 3) ^ (ml & 15)) * 16) + (nl & 4) * 2) = u32x2{pack16<F16>(o[0], o[1]), pack16<F16>(o[2], o[3])};
            }
        }
        __syncthreads();
        TL(3);
        bf16_t* const __restrict__ ob = a.out_bf16;
        const int chunk = tid & 15;
#pragma unroll
        for (int p0 = 0; p0 < 8; p0 += 4) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ml = (p0 + u) * 32 + (tid >> 4);
                v[u] = *(const u32x4*)(smem + ml * 256 + ((chunk ^ (ml & 15)) * 16));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = tile_m * G2_BM + (p0 + u) * 32 + (tid >> 4);
                if (m < a.M) store16_wt(ob + (size_t)m * a.ldo_bf16 + tile_n * 128 + chunk * 8, v[u]);
            }
        }
        TL(4);
        return;
    } else if constexpr (WN == 2 && ACT >= 0) {
        // bf16-only outputs (qkv, fc1): bias + activation in registers, the WHOLE tile staged as bf16 ([256 m][256 n], 512-byte
        // rows, 16-byte chunk index XOR (m & 31)) and written with 16 B per lane (two whole rows per wave-store): half the
        // store instructions and LDS bytes of the fp32 staging below -- 8-byte stores ran at ~4.7 B/clk/CU, store-issue-bound.
        if (fast && !vt_tile && a.out_bf16 && !a.out_f32 && !a.residual && (a.ldo_bf16 & 7) == 0 && ((uintptr_t)a.out_bf16 & 15) == 0) {
#if G2_DIRECT_EPI
            {
                // EXPERIMENT: straight from the accumulators, no LDS round trip and no block barrier.  One cross-lane swap per packed
                // register pair of two adjacent pieces leaves 8 consecutive n (16 B) of a row in each lane:
                //   16x16 layout (v_permlane16_swap: lane groups g = 0..3 of a row hold n = 4g..4g+3 of each 16-wide fragment):
                //     g0: frag nf n 0..7, g2: nf n 8..15, g1: nf+1 n 0..7, g3: nf+1 n 8..15  -> 64 contiguous bytes per row
                //   32x32 layout (v_permlane32_swap): lane l: piece p, lane l + 32: piece p + 1  -> 32 contiguous bytes per row
                const int nsel = M32 ? (lane >> 5) * 8 : (g & 1) * 16 + (g >> 1) * 8;
                bf16_t* const __restrict__ ob = a.out_bf16 + (size_t)tile_n * BN + wn * 128 + nsel;
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb) {
                    const int m = tile_m * G2_BM + wm * 64 + mb * MBS + lane_m;
                    bf16_t* const __restrict__ orow = ob + (size_t)m * a.ldo_bf16;
#pragma unroll
                    for (int p = 0; p < NP; p += 2) {
                        uint32_t pk[2][2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int nl = wn * 128 + (p + e) * PS + lane_n;
                            f32x4 b4 = {0, 0, 0, 0};
                            if (a.bias) b4 = *(const f32x4*)(a.bias + tile_n * BN + nl);
                            const f32x4 av = piece(p + e, mb);
                            float o[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float t = av[j] + b4[j];
                                if (ACT == SM_ACT_QUICK_GELU) t = t * sigmoidf_(1.702f * t);
                                o[j] = t;
                            }
                            pk[e][0] = pack16<F16>(o[0], o[1]);
                            pk[e][1] = pack16<F16>(o[2], o[3]);
                        }
                        u32x4 v;
                        if constexpr (M32) {
                            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                            v = u32x4{s0[0], s1[0], s0[1], s1[1]};
                        } else {
                            const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                            v = u32x4{s0[0], s1[0], s0[1], s1[1]};
                        }
                        if (m < a.M) {
#if G2_DIRECT_EPI == 2
                            *(u32x4*)(orow + p * PS) = v;
#else
                            store16_wt(orow + p * PS, v);
#endif
                        }
                    }
                }
                TL(3);
                TL(4);
                return;
            }
#endif
            // LayerNorm folding, consumer side (sm_linear_t.fold_stats_in): (-mu * rstd, rstd) of the lane's rows from the producer's partial sums
            f32x2 fst[NMB];
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) fst[mb] = f32x2{0.f, 1.f};
            const bool fold = a.fold_istats != nullptr;
            if (fold) {
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb) fst[mb] = fold_row_stats(a, tile_m * G2_BM + wm * 64 + mb * MBS + lane_m);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int nl = wn * 128 + p * PS + lane_n;
                f32x4 b4 = {0, 0, 0, 0}, g4 = {0, 0, 0, 0};
                if (fold) { b4 = *(const f32x4*)(a.fold_ic + tile_n * BN + nl); g4 = *(const f32x4*)(a.fold_ig + tile_n * BN + nl); }
                else if (a.bias) b4 = *(const f32x4*)(a.bias + tile_n * BN + nl);
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb) {
                    const int ml = wm * 64 + mb * MBS + lane_m;
                    const f32x4 av = piece(p, mb);
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = __builtin_fmaf(av[j], fst[mb][1], __builtin_fmaf(fst[mb][0], g4[j], b4[j]));      // fold off: av * 1 + (0 * 0 + b)
                        if (ACT == SM_ACT_QUICK_GELU) t = t * sigmoidf_(1.702f * t);
                        o[j] = t;
                    }
                    *(u32x2*)(smem + ml * 512 + (((nl >> 3) ^ (ml & 31)) * 16) + (nl & 4) * 2) = u32x2{pack16<F16>(o[0], o[1]), pack16<F16>(o[2], o[3])};
                }
            }
            __syncthreads();
            TL(3);
            bf16_t* const __restrict__ ob = a.out_bf16;
            const int chunk = tid & 31;
#pragma unroll
            for (int p0 = 0; p0 < 16; p0 += 4) {
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ml = (p0 + u) * 16 + (tid >> 5);
                    v[u] = *(const u32x4*)(smem + ml * 512 + ((chunk ^ (ml & 31)) * 16));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int m = tile_m * G2_BM + (p0 + u) * 16 + (tid >> 5);
                    if (m < a.M) {
                        if constexpr (ACT == SM_ACT_QUICK_GELU) store16_stream(ob + (size_t)m * a.ldo_bf16 + tile_n * BN + chunk * 8, v[u]);
                        else store16_wt(ob + (size_t)m * a.ldo_bf16 + tile_n * BN + chunk * 8, v[u]);
                    }
                }
            }
            TL(4);
            return;
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) { __syncthreads(); TL(3); }
        if ((wm >> 1) == half) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb) {
                    const int ml = (wm & 1) * 64 + mb * MBS + lane_m;
                    const int chunk = wn * 32 + (p * PS + lane_n) / 4;
                    *(f32x4*)(smem + ml * ROWB + ((chunk ^ (ml & 31)) * 16)) = piece(p, mb);
                }
        }
        __syncthreads();
        const int m0 = tile_m * G2_BM + half * 128;
        if (vt_tile) {
            const int nh = (a.N - a.vt_n0) / a.vt_dh;
            for (int pass = 0; pass < BN / NW; ++pass) {
                const int nl = pass * NW + wave;
                const int n = tile_n * BN + nl;
                if (n >= a.N) continue;
                const int c = n - a.vt_n0;
                const int h = c / a.vt_dh, d = c - h * a.vt_dh;
                const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int ml = lane + 64 * h2;
                    const int m = m0 + ml;
                    if (m < a.M) {
                        float v = *(const float*)(smem + ml * ROWB + (((nl >> 2) ^ (ml & 31)) * 16) + (nl & 3) * 4) + bv;
                        const int b = m / a.vt_S, sidx = m - b * a.vt_S;
                        a.vt[((size_t)(b * nh + h) * a.vt_dh + d) * a.vt_ld + sidx] = (bf16_t)cvt16<F16>(apply_act_rt(v, a.act));
                    }
                }
            }
        } else if (PFOLD) {             // (the launcher guarantees the fast path's conditions: whole column tiles, aligned rows)
            if constexpr (PFOLD)
                half_rows_epilogue_fold<F16>(smem, a.bias, a.residual, a.ldr, a.out_f32, a.ldo, a.out_bf16, a.ldo_bf16, a.fold_og, a.fold_ostats, tiles_n, tile_n, m0, a.M, tid);
        } else if (fast) {
            half_rows_epilogue<ACT, CPR, RPP, F16>(smem, a.bias, a.residual, a.ldr, a.out_f32, a.ldo, a.out_bf16, a.ldo_bf16, m0,
                                              tile_n * BN, a.M, tid);
        } else {
            for (int pass = 0; pass < 128 / RPP; ++pass) {
                const int ml = pass * RPP + tid / CPR;
                const int chunk = tid % CPR;
                f32x4 v = *(const f32x4*)(smem + ml * ROWB + ((chunk ^ (ml & 31)) * 16));
                store4(a, m0 + ml, tile_n * BN + chunk * 4, v, nullptr);
            }
        }
    }
    TL(4);
}

// ---------------------------------------------------------------------------------------------- persistent variant
// One 8-wave block per CU walks the tiles the dispatcher would have handed that CU one after the other (block b: XCD b % 8, slot
// b / 8; tiles slot, slot + 32, ... of the XCD's band), for bf16-only outputs with several tiles per CU (the ViT's QKV and fc1
// products: 3 and 4 rounds at 28 frames).  What it removes is the seam between two tiles of a CU -- in the one-tile-per-block
// kernel: epilogue 4.0-8.7 us with the matrix pipes idle, ~2.4 us until the next block's waves are up, 1.4 us until its first
// operands have crossed L2 (tools/gemm_timeline.hip) -- by keeping the k-loop's DMA pipeline running ACROSS the seam:
//   * the last three k-steps of a tile issue the first three k-steps of the NEXT tile (instead of nothing), the fourth follows as
//     soon as the last fragment reads of the old tile are done, so the next tile's k-loop starts with its operands in LDS;
//   * the epilogue therefore cannot use the ring: bias / activation / packing happen in registers, and the bf16 tile leaves through
//     a 32-KiB window behind the ring (LDS = 128 + 32 KiB, all of a CU's) in four passes of 256 rows x 64 columns (every wave
//     contributes 8 pieces per pass, so all four SIMDs write), 16-byte write-through stores;
//   * waits are counted by hand across the seam (vmcnt retires in order, the epilogue's stores sit between the prefetched stages
//     and the stages issued by the new tile's first k-steps): all four prefetched stages are waited for before the first store
//     (the register phase covers the fourth's flight), the new tile's k-steps 0..2 wait for nothing, k-step 3 waits for everything
//     older than its own two batches (the stores have had the rest of the epilogue and three k-steps to drain).
// The arithmetic is that of gemm256_kernel (same fragment order, same accumulation order): results are bit-identical.
// Every barrier is a raw s_barrier: __syncthreads() is an LDS fence, which hipcc turns into s_waitcnt vmcnt(0) while LDS-DMA is in
// flight -- exactly the wait this kernel exists to avoid.
template <int ACT, bool F16, bool FOLD = false>      // FOLD: consumer of a folded LayerNorm (a.fold_istats / fold_ig / fold_ic: sm_linear_t.fold_*)
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(LinArgs a, int tiles_m, int tiles_n, int cb) {
    constexpr int BN = 256;
    constexpr int STAGE = 32768, WINDOW = 4 * STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wn = wave >> 2, wm = wave & 3;
    const int KS = a.KS;
    const int nblk = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int q = nblk >> 3;                          // tiles per XCD band (the launcher guarantees nblk % 8 == 0)
    if (slot0 >= q) return;

    struct Tile { const char* wbase[2]; const char* xbase; uint32_t xoff[2]; int tile_m, tile_n; };
    auto locate = [&](int idx, Tile& t) {
        int tile_m, tile_n;
        if (cb > 0) {                                 // blocked walk inside the band, as gemm256_kernel
            const int rows_x = q / tiles_n;
            const int cg = idx / (rows_x * cb), rem = idx - cg * rows_x * cb;
            const int r = rem / cb;
            tile_m = xcd * rows_x + r;
            tile_n = cg * cb + (rem - r * cb);
        } else {
            const int bid = xcd * q + idx;
            tile_m = bid / tiles_n; tile_n = bid - tile_m * tiles_n;
        }
        t.tile_m = tile_m; t.tile_n = tile_n;
        t.xbase = (const char*)a.x + (size_t)tile_m * G2_BM * a.ldx * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int rgg = tile_n * 16 + wave * 2 + j;
            if constexpr (ACT == SM_ACT_SWIGLU_DUAL) rgg = ((wave * 2 + j) & 1) * (a.NRG >> 1) + tile_n * 8 + ((wave * 2 + j) >> 1);      // gate / up groups alternate (gemm256_kernel)
            if (rgg >= a.NRG) rgg = a.NRG - 1;
            t.wbase[j] = (const char*)a.w + (size_t)rgg * KS * 1024;
            const int row = (wave * 2 + j) * 16 + (lane >> 2);
            int rl = row;
            if (tile_m * G2_BM + row >= a.M) rl = a.M - 1 - tile_m * G2_BM;
            const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
            t.xoff[j] = (uint32_t)(rl * a.ldx + chunk * 8) * 2;
        }
    };
    const uint32_t woff = lane * 16;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    auto dma = [&](const char* sbase, uint32_t voff, uint32_t dst) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(dst) : "memory");
    };
    auto stage = [&](const Tile& t, int ks, int slot) {
        const uint32_t sb = lds0 + slot * STAGE + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) dma(t.wbase[j] + (size_t)ks * 1024, woff, sb + j * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) dma(t.xbase + (size_t)ks * 64, t.xoff[j], sb + 16384 + j * 1024);
    };

    Tile cur, nxt;
    int idx = slot0;
    locate(idx, cur);
    nxt = cur;
    // prologue = the state behind a seam: stages 0..3 landed (the tile body then needs no "first tile" variant --
    // a branch inside the body makes hipcc shuffle the 128 accumulators at the join, 130+ spills)
    stage(cur, 0, 0);
    stage(cur, 1, 1);
    stage(cur, 2, 2);
    stage(cur, 3, 3);
    // LayerNorm folding: (-mu * rstd, rstd) of this lane's four rows of the CURRENT tile travel through the k-loop in 8 registers; the next tile's are
    // fetched at the START of this tile's register phase (their flight hides behind its arithmetic) and finalised behind the wait that phase ends with.
    // Fetching them where they are used cost 2.3 us per tile in situ (q|k|v 87 -> 94 us, fc1 125 -> 135 us): the partial sums were written by the kernel
    // before this one, from other XCDs -- every first touch is a miss, and the whole register phase waits for it.
    // A lane's four rows (mf * 16 + i) are also the rows of the three other lane groups of its wave: lane (g, i) fetches and finalises ROW g * 16 + i only
    // and the four groups swap results (8 ds_bpermute) -- 8 registers in flight instead of 32 (with 32 the epilogue spilled a Tile register across the
    // tile loop, and the reload's compiler-inserted vmcnt(0) at the head of the next tile waited for the epilogue's stores).
    f32x2 fst[4];
    if constexpr (FOLD) {
        const f32x2 own = fold_row_stats(a, cur.tile_m * G2_BM + wm * 64 + g * 16 + i);
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) fst[mf] = f32x2{__shfl(own[0], mf * 16 + i, 64), __shfl(own[1], mf * 16 + i, 64)};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x4 acc[8][4];
    // MODE: what the R interval of a k-step issues and waits for
    //   0 steady: stage ks + 3 of this tile, leave two batches in flight      1 / 2 last tile's tail: nothing to issue, vmcnt(4) / (0)
    //   3 the tile's last three k-steps when another tile follows: stage ks + 3 - KS of the NEXT tile, two batches in flight
    //   4 / 5 k-steps 0 / 1, 2 behind a seam: nothing / stage ks + 3 to issue, NO wait (stages 1..3 were waited for before the
    //     epilogue's stores were issued; waiting on a count here would wait for the stores)
    auto kstep = [&](int ks, auto mode, int slot) {
        constexpr int MODE = decltype(mode)::value;
        const char* sw = smem + slot * STAGE;
        const char* sx = sw + 16384;
        bf16x8 xf[4], wf[8];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int ml = wm * 64 + mf * 16 + i;
            xf[mf] = *(const bf16x8*)(sx + ml * 64 + ((g ^ ((0 - (ml >> 2)) & 3)) * 16));
        }
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) wf[nf] = *(const bf16x8*)(sw + (wn * 8 + nf) * 1024 + lane * 16);
        if (MODE == 0) {
            stage(cur, ks + 3, (slot + 3) & 3);
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        } else if (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        } else if (MODE == 2) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else if (MODE == 3) {
            stage(nxt, ks + 3 - KS, (slot + 3) & 3);
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        } else if (MODE == 4) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            stage(cur, ks + 3, (slot + 3) & 3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nf = 0; nf < 8; ++nf)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
                acc[nf][mf] = mfma16<F16>(wf[nf], xf[mf], acc[nf][mf]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using M0 = std::integral_constant<int, 0>; using M3 = std::integral_constant<int, 3>;
    using M4 = std::integral_constant<int, 4>; using M5 = std::integral_constant<int, 5>;

    while (true) {
        const int nidx = idx + nslots;
        const bool has_next = nidx < q;
        locate(has_next ? nidx : idx, nxt);            // behind the last tile the "next" stages re-fetch this tile's (never read): no tail variant
#pragma unroll
        for (int nf = 0; nf < 8; ++nf)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) acc[nf][mf] = f32x4{0, 0, 0, 0};
        if (wn == 1) __builtin_amdgcn_s_barrier();                  // the two wave groups run one barrier apart (gemm256_kernel)
        // ---- k-loop (KS % 4 == 0, KS >= 8: ring slot = ks & 3 at compile time)
        kstep(0, M4{}, 0);
        kstep(1, M5{}, 1);
        kstep(2, M5{}, 2);
        kstep(3, M0{}, 3);
        int ks = 4;
        for (; ks + 4 < KS; ks += 4) {
            kstep(ks, M0{}, 0); kstep(ks + 1, M0{}, 1); kstep(ks + 2, M0{}, 2); kstep(ks + 3, M0{}, 3);
        }
        kstep(ks, M0{}, 0);
        kstep(ks + 1, M3{}, 1); kstep(ks + 2, M3{}, 2); kstep(ks + 3, M3{}, 3);
        if (wn == 0) __builtin_amdgcn_s_barrier();                  // realign: every wave is done with the ring's last stage
        // the next tile's fourth stage goes into the slot the old tile's last k-step was read from -- issued BEFORE the register phase,
        // which then covers its flight: behind it all four prefetched stages are waited for and no store has been issued yet
        stage(nxt, 3, 3);
        // ---- epilogue, register phase: bias, activation, packing (all waves at once: four SIMDs).
        // Every per-lane address of the epilogue is derived from an OPAQUE copy of the thread id, re-made per tile: from the plain
        // one hipcc hoists ~60 tile-invariant address registers out of the tile loop and keeps them alive through the k-loop, whose
        // 128 accumulators + 48 fragment registers leave no room for them (147 spills).
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        const int lane_e = tid_e & 63, i_e = lane_e & 15, g_e = lane_e >> 4;
        constexpr bool DUAL = ACT == SM_ACT_SWIGLU_DUAL;      // SwiGLU-dual: fragments 2f / 2f + 1 are gate / up of the same 16 columns -> 4 output fragments per wave
        constexpr int NPK = DUAL ? 4 : 8, OBN = DUAL ? 128 : 256;
        uint32_t pk[NPK][4][2];
        f32x4 fraw[2];                 // FOLD: the next tile's partial row sums of this lane's share (one row)
        f32x2 fst_next[4];
        if constexpr (DUAL) {
            const int F = a.N >> 1;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int nl = wn * 64 + f * 16 + g_e * 4;
                f32x4 bg = {0, 0, 0, 0}, bu = {0, 0, 0, 0};
                if (a.bias) { bg = *(const f32x4*)(a.bias + cur.tile_n * 128 + nl); bu = *(const f32x4*)(a.bias + F + cur.tile_n * 128 + nl); }
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = siluf_(acc[2 * f][mf][j] + bg[j]) * (acc[2 * f + 1][mf][j] + bu[j]);
                    pk[f][mf][0] = pack16<F16>(o[0], o[1]);
                    pk[f][mf][1] = pack16<F16>(o[2], o[3]);
                }
            }
        } else {
        // LayerNorm folding: the NEXT tile's partial sums (plain loads, issued behind the next tile's fourth stage like the bias loads: hipcc counts its own
        // loads only, so its waits cover the older DMA as well -- over-waiting, never early); consumed together with the column vectors below
        if constexpr (FOLD) {
            int mrow = nxt.tile_m * G2_BM + wm * 64 + g_e * 16 + i_e;          // this lane's share: the row of ITS lane group (see the prologue)
            if (mrow >= a.M) mrow = a.M - 1;
            const f32x4* sp = (const f32x4*)(a.fold_istats + (size_t)mrow * 8);
            fraw[0] = sp[0]; fraw[1] = sp[1];
        }
        // (FOLD) all 16 column vectors are fetched up front and PINNED as landed (the empty asm reads them, so hipcc waits for ALL its loads there): hipcc sinks the arithmetic of the later window passes
        // between the passes (good: it runs under the stores' flight), and with the loads left where they are used its counted waits -- it does not see
        // the asm stores in the queue -- waited for every earlier pass's write-through stores to complete: +2 us per tile, in isolation and in situ
        f32x4 fgv[8], fcv[8];
        if constexpr (FOLD) {
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) {
                const int nl = wn * 128 + nf * 16 + g_e * 4;
                fcv[nf] = *(const f32x4*)(a.fold_ic + cur.tile_n * BN + nl);
                fgv[nf] = *(const f32x4*)(a.fold_ig + cur.tile_n * BN + nl);
            }
            asm volatile("" : "+v"(fgv[0]), "+v"(fgv[1]), "+v"(fgv[2]), "+v"(fgv[3]), "+v"(fgv[4]), "+v"(fgv[5]), "+v"(fgv[6]), "+v"(fgv[7]),
                              "+v"(fcv[0]), "+v"(fcv[1]), "+v"(fcv[2]), "+v"(fcv[3]), "+v"(fcv[4]), "+v"(fcv[5]), "+v"(fcv[6]), "+v"(fcv[7]));
            // the next tile's (-mu * rstd, rstd), finalised at once (fold_row_stats' arithmetic; the 32 raw registers are free again before the tile's own arithmetic starts)
            {
                const f32x4 u = fraw[0], v = fraw[1];
                const float s1 = ((u[0] + u[2]) + v[0]) + v[2], s2 = ((u[1] + u[3]) + v[1]) + v[3];
                const float mu = s1 * a.fold_invd;
                const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(s2, a.fold_invd, a.fold_eps) - mu * mu);
                const float o0 = -mu * rstd;
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) fst_next[mf] = f32x2{__shfl(o0, mf * 16 + i_e, 64), __shfl(rstd, mf * 16 + i_e, 64)};
            }
        }
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) {
            const int nl = wn * 128 + nf * 16 + g_e * 4;
            f32x4 b4 = {0, 0, 0, 0}, g4 = {0, 0, 0, 0};
            if constexpr (FOLD) { b4 = fcv[nf]; g4 = fgv[nf]; }
            else if (a.bias) b4 = *(const f32x4*)(a.bias + cur.tile_n * BN + nl);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t;
                    if constexpr (FOLD) t = __builtin_fmaf(acc[nf][mf][j], fst[mf][1], __builtin_fmaf(fst[mf][0], g4[j], b4[j]));
                    else t = acc[nf][mf][j] + b4[j];
                    if (ACT == SM_ACT_QUICK_GELU) t = t * sigmoidf_(1.702f * t);
                    o[j] = t;
                }
                pk[nf][mf][0] = pack16<F16>(o[0], o[1]);
                pk[nf][mf][1] = pack16<F16>(o[2], o[3]);
            }
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (FOLD) {
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) fst[mf] = fst_next[mf];          // (the register phase above was the last reader of this tile's)
        }
        // ---- window passes: pass qn moves columns [32 qn, 32 qn + 32) of both 128-column halves (fragments 2 qn, 2 qn + 1 of
        // every wave).  Window row = 128 B (8 chunks of 16 B: 4 of the wn = 0 half, 4 of the wn = 1 half), chunk index XOR
        // ((row >> 1) & 7): the 16 lanes of a ds_write_b64 group meet 16 distinct (row parity, chunk) bank groups.
        char* const win = smem + WINDOW;
        bf16_t* const __restrict__ ob = a.out_bf16;
#pragma unroll
        for (int qn = 0; qn < NPK / 2; ++qn) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {
                    const int row = wm * 64 + mf * 16 + i_e;
                    const int chunk = wn * 4 + e * 2 + (g_e >> 1);
                    *(u32x2*)(win + row * 128 + ((chunk ^ ((row >> 1) & 7)) * 16) + (g_e & 1) * 8) = u32x2{pk[qn * 2 + e][mf][0], pk[qn * 2 + e][mf][1]};
                }
            // (the s_barrier builtin is no memory operation to the optimiser: without the compiler fences on BOTH sides it moved window
            // reads in front of the barrier -- 1 in 40 000 outputs stale)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int id = j * 512 + tid_e;
                const int row = id >> 3, c = id & 7;
                const u32x4 v = *(const u32x4*)(win + row * 128 + ((c ^ ((row >> 1) & 7)) * 16));
                const int m = cur.tile_m * G2_BM + row;
                const int n = cur.tile_n * OBN + (c >> 2) * (OBN / 2) + qn * 32 + (c & 3) * 8;
                if (m < a.M) {
                    if constexpr (ACT == SM_ACT_QUICK_GELU) store16_stream(ob + (size_t)m * a.ldo_bf16 + n, v);
                    else store16_wt(ob + (size_t)m * a.ldo_bf16 + n, v);
                }
            }
            if (qn < NPK / 2 - 1) {                                           // the window is free again (the reads were consumed by the stores)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        if (!has_next) break;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur = nxt; idx = nidx;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the dummy stages behind the last tile must not outlive the block's LDS
}

template <int WN, bool F16>
static int launch_wn(const LinArgs& a, int act, hipStream_t st, bool allow_persistent = true, int S = 1) {
    constexpr int BN = 128 * WN;
    constexpr int LDS = WN == 2 ? 4 * 32768 : 3 * (BN * 64 + 16384);
    const bool dual = act == SM_ACT_SWIGLU_DUAL;                 // tiles of 256 weight rows = 128 gate + 128 up rows -> 128 output columns (WN == 2 only: launch_gemm256 checks)
    const int tiles_m = cdiv(a.M, G2_BM), tiles_n = dual ? (a.N >> 1) / 128 : cdiv(a.N, BN);
    static bool attr_set = false;
    if (!attr_set) {
        if constexpr (WN == 2) SM_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<SM_ACT_SWIGLU_DUAL, WN, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        SM_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<0, WN, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        SM_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<1, WN, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        SM_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<-1, WN, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        if constexpr (WN == 2) SM_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<0, WN, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const dim3 grid(tiles_m * tiles_n, S);              // S > 1 (WN == 2, one-tile kernel only): split-K slabs, see the kernel
    if (S > 1) allow_persistent = false;
    // blocked tile walk: the XCD's band is walked in SM_GEMM_CG column groups (0 = plain row-major walk; default 3, or the next smaller count that divides the column tiles); needs whole
    // bands per XCD and a group count that divides the column tiles
    static int cg_env = -1;
    if (cg_env < 0) { const char* e = getenv("SM_GEMM_CG"); cg_env = e ? atoi(e) : 3; }
    const int nblk = tiles_m * tiles_n;
    int cb = 0;
    if (cg_env > 1 && (nblk & 7) == 0 && (nblk >> 3) % tiles_n == 0) {
        int g = cg_env;
        while (g > 1 && tiles_n % g) --g;
        if (g > 1) cb = tiles_n / g;
    }
    if constexpr (WN == 2) {
        // persistent variant: bf16-only full tiles, several tiles per CU, whole XCD bands (SM_GEMM_PERSIST=0 switches it off)
        static int persist = -1, n_cu = 0;
        if (persist < 0) {
            const char* e = getenv("SM_GEMM_PERSIST");
            persist = e ? atoi(e) : 1;
            int dev = 0; hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
            static bool attr_p = false;
            if (!attr_p) {
                SM_HIP(hipFuncSetAttribute((const void*)gemm256p_kernel<0, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768));
                SM_HIP(hipFuncSetAttribute((const void*)gemm256p_kernel<1, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768));
                SM_HIP(hipFuncSetAttribute((const void*)gemm256p_kernel<SM_ACT_SWIGLU_DUAL, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768));
                SM_HIP(hipFuncSetAttribute((const void*)gemm256p_kernel<0, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768));
                SM_HIP(hipFuncSetAttribute((const void*)gemm256p_kernel<1, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768));
                attr_p = true;
            }
        }
        const bool bf16_only = a.out_bf16 && !a.out_f32 && !a.residual && !a.vt && a.remap_in == 0 && (a.ldo_bf16 & 7) == 0 &&
                               ((uintptr_t)a.out_bf16 & 15) == 0 && (a.N % BN) == 0 && (act == SM_ACT_NONE || act == SM_ACT_QUICK_GELU || dual);
        if (persist && allow_persistent && bf16_only && (!a.fold_istats || a.fold_itiles == 4) && n_cu >= 8 && (n_cu & 7) == 0 && (nblk & 7) == 0 && nblk > n_cu && (a.KS & 3) == 0 && a.KS >= 8 &&
            (cb == 0 || (nblk >> 3) % tiles_n == 0)) {
            const dim3 pgrid(n_cu);
            if (a.fold_istats && !dual) {
                if (act == SM_ACT_NONE) gemm256p_kernel<0, F16, true><<<pgrid, 512, 5 * 32768, st>>>(a, tiles_m, tiles_n, cb);
                else gemm256p_kernel<1, F16, true><<<pgrid, 512, 5 * 32768, st>>>(a, tiles_m, tiles_n, cb);
            } else if (act == SM_ACT_NONE) gemm256p_kernel<0, F16><<<pgrid, 512, 5 * 32768, st>>>(a, tiles_m, tiles_n, cb);
            else if (dual) gemm256p_kernel<SM_ACT_SWIGLU_DUAL, F16><<<pgrid, 512, 5 * 32768, st>>>(a, tiles_m, tiles_n, cb);
            else gemm256p_kernel<1, F16><<<pgrid, 512, 5 * 32768, st>>>(a, tiles_m, tiles_n, cb);
            SM_LAUNCH_CHECK();
            return SM_OK;
        }
    }
    if constexpr (WN == 2) {
        if (dual) {
            gemm256_kernel<SM_ACT_SWIGLU_DUAL, WN, F16><<<grid, 256 * WN, LDS, st>>>(a, tiles_m, tiles_n, cb);
            SM_LAUNCH_CHECK();
            return SM_OK;
        }
    }
    if (a.fold_ostats) {
        // producer of a folded LayerNorm: the one-tile kernel's fp32 epilogue in its own instantiation (sm_linear checked the rest)
        if constexpr (WN == 2) {
            if (act != SM_ACT_NONE || S > 1 || (a.N % BN) || a.vt || a.remap_in || !a.out_f32 || !a.out_bf16 || !a.fold_og) return SM_EINVAL;
            gemm256_kernel<0, WN, F16, true><<<grid, 256 * WN, LDS, st>>>(a, tiles_m, tiles_n, cb);
            SM_LAUNCH_CHECK();
            return SM_OK;
        } else return SM_EINVAL;
    }
    if (act == SM_ACT_NONE) gemm256_kernel<0, WN, F16><<<grid, 256 * WN, LDS, st>>>(a, tiles_m, tiles_n, cb);
    else if (act == SM_ACT_QUICK_GELU) gemm256_kernel<1, WN, F16><<<grid, 256 * WN, LDS, st>>>(a, tiles_m, tiles_n, cb);
    else gemm256_kernel<-1, WN, F16><<<grid, 256 * WN, LDS, st>>>(a, tiles_m, tiles_n, cb);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// bn = 256: one 8-wave block per CU; bn = 128: two independent 4-wave blocks per CU (their phases drift apart, so one
// block's barriers / epilogue overlap the other's MFMAs)
int launch_gemm256(const LinArgs& a, int act, int bn, hipStream_t st, int S) {
    const bool np = bn == 257;                    // SM_TILE_256_ONE_TILE_PER_BLOCK: never the persistent variant (tests, A/B)
    if (S > 1 && (bn == 128 || act != SM_ACT_NONE || !a.out_f32 || a.out_bf16 || a.residual || a.bias || a.vt || a.remap_in || a.KS % S)) return SM_EINVAL;
    if (a.f16) return bn == 128 ? launch_wn<1, true>(a, act, st) : launch_wn<2, true>(a, act, st, !np, S);
    return bn == 128 ? launch_wn<1, false>(a, act, st) : launch_wn<2, false>(a, act, st, !np, S);
}
