// Weight-streaming MFMA product for 33..128 activation rows (round 5): a batched decode step of 33..128 streams (sm_group_llm_decode: M = streams).
//
// At these row counts a linear is a weight stream (Mistral-7B: 14.2 GB per step) with 64..256 FLOP per weight byte: HBM-bound as long as the weights
// keep coming.  The 128 x 128 tiled GEMM these calls used stages W and X through one LDS ring, so at most three 16-KiB W stages (48 KiB per CU) are in
// flight: 3.0-4.1 TB/s on the four products of a layer (profiles/r05_group_decode128_step_timeline.txt).  Here the two operands take different roads:
//   * W never touches LDS.  A wave owns TWO 16-row groups of the packed weight image; one 16-byte load per lane is one MFMA A operand (fragment-major
//     layout, 1 KiB contiguous per wave-load).  Every consumer wave keeps a register ring of DW = 8 k-steps (16 KiB) in flight -- plain loads, so the
//     compiler's in-order vmcnt counting is exact and nothing younger is ever waited for: 64 KiB of weights in flight per 4-wave block, two blocks per CU.
//   * X (<= 128 rows x 32 k per k-step = 8 KiB) goes through an 8-stage LDS ring filled by TWO MORE waves with LDS-DMA (global_load_lds): their loads hit L2,
//     and because vmcnt retires in order PER WAVE they must not share a queue with the HBM loads of the weight ring -- a consumer that waited for its X
//     rows would drain its whole W ring first.  The loader counts its own waits by hand; one raw s_barrier per k-step hands a stage over.
//   Consumer wave per k-step: 8 B-fragment reads (ds_read_b128, the XOR-swizzled [row][32 k] image of gemm256.hip), 2 x 8 MFMAs 16x16x32, refill of its ring.
// Grid = (column blocks of 128 weight rows, S K-slabs); S > 1: raw accumulators into fp32 slabs [S][M][N] (summed, with the epilogue and an optional
// RMSNorm / LayerNorm of the finished row, by linear.hip's slab passes); S == 1: the epilogue here (store4).  SwiGLU-dual (SM_ACT_SWIGLU_DUAL): a wave's two
// row groups are the gate and the up group of the SAME 16 output columns, act_fn(gate) * up in the lane, 16-bit rows out.
#include <type_traits>

#include "linear_common.h"

// experiment switches of tools/experiments/wstream_probe.hip (all 0 in the product): WS_X_NOMFMA: no MFMAs (pure streaming), WS_X_NOXREAD: no B-fragment reads,
// WS_X_NOBAR: no per-k-step barriers (wrong results; what the hand-over costs)
#ifndef WS_X_NOMFMA
#define WS_X_NOMFMA 0
#endif
#ifndef WS_X_NOXREAD
#define WS_X_NOXREAD 0
#endif
#ifndef WS_X_NOBAR
#define WS_X_NOBAR 0
#endif
#ifndef WS_DW
#define WS_DW 8          // k-steps of weights in flight per consumer wave
#endif
#define WS_PF 4          // k-steps of X in flight per loader wave (registers)
#ifndef WS_NST
#define WS_NST 4         // LDS stages of X (one 32-deep k-step each)
#endif

template <bool F16, bool DUAL>
__global__ __launch_bounds__(384, 2) void wstream_kernel(LinArgs a, float* __restrict__ ws, int ksl) {
    __shared__ __attribute__((aligned(16))) char sx[WS_NST * 8192];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int KS = a.KS;
    const int ks0 = blockIdx.y * ksl;
    const int nks = KS - ks0 < ksl ? KS - ks0 : ksl;           // (the launcher makes every slab a multiple of WS_DW k-steps)

    if (wave >= 4) {
        // ---- two loader waves: X rows of k-step t into stage t % NST, four 16-row pieces each.  One piece = 16 rows x 64 B (lane >> 2 = row, lane & 3 =
        // 16-byte chunk; the chunk index is XOR-swizzled on the SOURCE address: slot c of row r holds k-chunk c ^ P(r), P(r) = (-(r >> 2)) & 3).
        // Register-staged (global_load_dwordx4 -> ds_write_b128, a ring of WS_PF k-steps in flight), not LDS-DMA: a global_load_lds costs ~100-150 clk of
        // issue (M0 + per-lane 64-bit address), eight per k-step made the LOADERS the pace of the block (0.42 us per k-step whatever the ring depths);
        // these waves' vmcnt queue holds only X loads (L2 hits), so hipcc's in-order counting is exact here too.
        const int j0 = (wave - 4) * 4;
        const u32x4* src[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (j0 + j) * 16 + (lane >> 2);
            const int rl = row < a.M ? row : a.M - 1;
            const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
            src[j] = (const u32x4*)((const char*)a.x + ((size_t)rl * a.ldx + (size_t)ks0 * 32 + chunk * 8) * 2);
        }
        u32x4 xr[WS_PF][4];
#pragma unroll
        for (int d = 0; d < WS_PF; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[d][j] = src[j][(size_t)d * 4];          // k-step d: + 64 B
            __builtin_amdgcn_sched_barrier(0);
        }
        auto put = [&](int t, const u32x4 (&x)[4]) {
            char* dst = sx + (t & (WS_NST - 1)) * 8192 + j0 * 1024 + lane * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) *(u32x4*)(dst + j * 1024) = x[j];
        };
        put(0, xr[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[0][j] = src[j][(size_t)WS_PF * 4];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // stage 0 is in LDS
        // iteration t (behind the barrier that published stage t): write k-step t + 1 (ring slot (t + 1) % PF) into stage (t + 1) % NST -- last read during
        // k-step t + 1 - NST, long closed --, refill the slot with k-step t + 1 + PF, barrier (it publishes stage t + 1).  Trips of WS_PF iterations with
        // compile-time slots; STEADY: every refill of the trip is in range (no run-time branch: see the consumers' trip), else checked per iteration.
        auto ltrip = [&](int t0, auto steady) {
            constexpr bool STEADY = decltype(steady)::value;
#pragma unroll
            for (int d = 0; d < WS_PF; ++d) {
                const int t = t0 + d;
                constexpr int sl = 0; (void)sl;
                if (STEADY || t + 1 < nks) put(t + 1, xr[(d + 1) % WS_PF]);
                if (STEADY || t + 1 + WS_PF < nks) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xr[(d + 1) % WS_PF][j] = src[j][(size_t)(t + 1 + WS_PF) * 4];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!WS_X_NOBAR) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int t0 = 0;
        for (; t0 + 2 * WS_PF < nks; t0 += WS_PF) ltrip(t0, std::true_type{});
        for (; t0 < nks; t0 += WS_PF) ltrip(t0, std::false_type{});       // (nks % WS_PF == 0)
        return;
    }

    // ---- consumers
    int rgA, rgB;
    if (DUAL) { rgA = blockIdx.x * 4 + wave; rgB = (a.NRG >> 1) + rgA; }
    else { rgA = blockIdx.x * 8 + wave * 2; rgB = rgA + 1; }
    const bool okA = DUAL ? rgA < (a.NRG >> 1) : rgA < a.NRG, okB = DUAL ? okA : rgB < a.NRG;
    const bf16x8* wpA = a.w + ((size_t)(okA ? rgA : 0) * KS + ks0) * 64 + lane;
    const bf16x8* wpB = a.w + ((size_t)(okB ? rgB : 0) * KS + ks0) * 64 + lane;
    bf16x8 wr[WS_DW][2];
#pragma unroll
    for (int d = 0; d < WS_DW; ++d) {                    // (nks >= WS_DW)
        // in the ORDER the trips refill (A, B per k-step): hipcc sorted these sixteen loads by address (all A, then all B), and its wait-count pass merges that
        // queue order into the loop's -- slot 2 was then waited for with vmcnt(5) instead of 14, i.e. the ring ran three k-steps deep instead of eight
        wr[d][0] = __builtin_nontemporal_load(wpA + (size_t)d * 64);
        wr[d][1] = __builtin_nontemporal_load(wpB + (size_t)d * 64);
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 acc[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) acc[r][mf] = f32x4{0, 0, 0, 0};
    __builtin_amdgcn_s_barrier();                        // stage 0 is in LDS
    // Software pipeline: the B fragments of k-step t are READ (8 ds_read_b128, all issued together) right behind the barrier that publishes stage t, and
    // MULTIPLIED one k-step later, behind the next barrier -- the LDS round trip hides under the previous k-step's 16 MFMAs instead of being paid four
    // times per k-step (two reads, wait, four MFMAs: what hipcc schedules when reads and MFMAs of one k-step sit in one block: ~1000 clk per k-step).
    // one trip = WS_DW k-steps with compile-time ring slots; REFILL is a compile-time switch: with a run-time branch around the refill hipcc's wait-count
    // analysis gives up at the loop's back edge and drains the whole ring (vmcnt(0)) once per trip
    bf16x8 xf[2][8];
    auto read_stage = [&](int t, bf16x8 (&x)[8]) {
        const char* st = sx + (t & (WS_NST - 1)) * 8192;
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) {
            const int ml = mf * 16 + i;
            x[mf] = *(const bf16x8*)(st + ml * 64 + ((g ^ ((0 - (ml >> 2)) & 3)) * 16));
        }
    };
    read_stage(0, xf[0]);
    auto trip = [&](int t0, auto refill, auto last) {
        constexpr bool REFILL = decltype(refill)::value, LAST = decltype(last)::value;
#pragma unroll
        for (int d = 0; d < WS_DW; ++d) {
            const int t = t0 + d;
            // every wave's reads of stage t have returned -> the loaders may refill it; stage t + 1 is published by the same barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!WS_X_NOBAR) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!WS_X_NOXREAD && !(LAST && d == WS_DW - 1)) read_stage(t + 1, xf[(d + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 w0 = wr[d][0], w1 = wr[d][1];
#pragma unroll
            for (int mf = 0; mf < 8; ++mf) {
                if (WS_X_NOMFMA) { if (mf == 0) { acc[0][0] += __builtin_bit_cast(f32x4, w0); acc[1][0] += __builtin_bit_cast(f32x4, w1); } continue; }
                acc[0][mf] = mfma16<F16>(w0, xf[d & 1][mf], acc[0][mf]);
                acc[1][mf] = mfma16<F16>(w1, xf[d & 1][mf], acc[1][mf]);
            }
            if (REFILL) {
                wr[d][0] = __builtin_nontemporal_load(wpA + (size_t)(t + WS_DW) * 64);
                wr[d][1] = __builtin_nontemporal_load(wpB + (size_t)(t + WS_DW) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int t0 = 0;
    for (; t0 + 2 * WS_DW <= nks; t0 += WS_DW) trip(t0, std::true_type{}, std::false_type{});
    trip(t0, std::false_type{}, std::true_type{});       // (nks % WS_DW == 0, nks >= WS_DW: the last trip refills nothing)

    // ---- epilogue.  acc[r][mf]: lane (g, i) holds n = rg_r * 16 + g * 4 + 0..3 of row m = mf * 16 + i
    if (DUAL) {
        if (!okA) return;
        const int F = a.N >> 1;
        const int n0 = rgA * 16 + g * 4;
        f32x4 bg = {0, 0, 0, 0}, bu = {0, 0, 0, 0};
        if (a.bias) { bg = *(const f32x4*)(a.bias + n0); bu = *(const f32x4*)(a.bias + F + n0); }
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) {
            const int m = mf * 16 + i;
            if (m < a.M) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = siluf_(acc[0][mf][j] + bg[j]) * (acc[1][mf][j] + bu[j]);
                *(u32x2*)(a.out_bf16 + (size_t)m * a.ldo_bf16 + n0) = u32x2{pack16<F16>(o[0], o[1]), pack16<F16>(o[2], o[3])};
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (!(r ? okB : okA)) continue;
        const int n0 = (r ? rgB : rgA) * 16 + g * 4;
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) {
            const int m = mf * 16 + i;
            if (m >= a.M) continue;
            if (ws) {
                if (n0 + 3 < a.N) *(f32x4*)(ws + ((size_t)blockIdx.y * a.M + m) * a.N + n0) = acc[r][mf];
                else
                    for (int j = 0; j < 4; ++j)
                        if (n0 + j < a.N) ws[((size_t)blockIdx.y * a.M + m) * a.N + n0 + j] = acc[r][mf][j];
            } else {
                store4(a, m, n0, acc[r][mf], nullptr);
            }
        }
    }
}

// S = 1: epilogue in the kernel; S > 1: slabs in `ws` ([S][M][N] fp32, the caller sums them).  ksl = k-steps per slab (a multiple of WS_DW).
int launch_wstream(const LinArgs& a, hipStream_t st, float* ws, int S, int ksl) {
    const bool dual = a.act == SM_ACT_SWIGLU_DUAL;
    const dim3 grid(dual ? cdiv(a.NRG >> 1, 4) : cdiv(a.NRG, 8), S);
    if (a.f16) {
        if (dual) wstream_kernel<true, true><<<grid, 384, 0, st>>>(a, ws, ksl);
        else wstream_kernel<true, false><<<grid, 384, 0, st>>>(a, ws, ksl);
    } else {
        if (dual) wstream_kernel<false, true><<<grid, 384, 0, st>>>(a, ws, ksl);
        else wstream_kernel<false, false><<<grid, 384, 0, st>>>(a, ws, ksl);
    }
    SM_LAUNCH_CHECK();
    return SM_OK;
}
