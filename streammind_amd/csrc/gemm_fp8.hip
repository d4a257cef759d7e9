// fp8 x fp8 GEMM on the CDNA4 block-scaled matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4, twice the bf16 MFMA rate) for
// calls with MORE than 16 activation rows in the opt-in fp8 mode (BASELINE configs[4]: LLM prefill chunks, teacher-forced
// evaluation; `weights_fp8 = 2` / SM_W_FP8_MFMA).  Up to 16 rows the fp8 weights are streamed once and expanded in registers
// (linear.hip: HBM-bound, bf16 activations); above that the round-2 build expanded the whole fp8 image to a bf16 scratch per call
// and ran the bf16 GEMM.  Here both operands are fp8:
//   W  the sm_quant_pack_weight_fp8 image as it is: e4m3, per-output-row scale sw[n]; [N/16][K/64][lane][16 B], a lane (g, i)
//      holding W[rg*16 + i][64c + 8g + 0..7] and [64c + 32 + 8g + 0..7] of chunk c
//   X  quantised per ROW (per token) to e4m3 by quant_rows_fp8_kernel: sx[m] = max|x[m,:]| / 448, q = fp8(x / sx[m]) (RNE, the
//      same rule as the weights) and written fragment-major: [M/16][K/128][2][lane][16 B]
//   y[m][n] = sx[m] * sw[n] * sum_k qx[m][k] * qw[n][k]   (fp8 products are exact in fp32, fp32 accumulation) + bias, act, residual
// The instruction multiplies a 16 x 128 by a 128 x 16 fragment; lane group g supplies "its" 32 k of a row.  WHICH 32 of the 128 is
// free as long as A and B agree (the sum over k does not care), so a lane's 32 operand bytes are simply its 16 bytes of weight
// chunk 2q followed by its 16 bytes of chunk 2q + 1 -- the fp8 image needs no repacking -- and the activation quantiser writes X
// in exactly that k order.  The block scales of the MX format are all 1 (e8m0 127): the per-row scales are applied in fp32 on
// the way out.
// Tile 128(n) x BM(m) x 128(k) per 4-wave block (BM = 128, or 64 when M is small: more blocks, less padding), a ring of three LDS
// stages of (16 + BM/8) KiB filled by global_load_lds TWO k-blocks ahead (a k-block is only 8-16 MFMAs per wave: one block of
// cover does not hide an L2 round trip), counted waits, one barrier per k-block; a wave owns 64(n) x BM/2(m).
#include <stdlib.h>
#include <map>
#include <mutex>

#include "linear_common.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;

// ---------------------------------------------------------------------------------------------- activation quantiser
// grid (16-row groups, column slices): every block finds the row maxima of its 16 rows (wave w: rows 4w..4w+3; the slices of a
// row group each read the rows once more -- L2 hits -- so that a 300-row call is 4 x 19 blocks instead of 19), then writes the
// 16-byte fragment pieces of its slice of k-blocks.  x: 16-bit (bf16 or fp16) [M][ldx]; rows >= M quantise to zero.  K % 128 == 0.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, int M, int K, int ldx, int f16,
                                                             u32x4* __restrict__ xq, float* __restrict__ xscale) {
    __shared__ float s_inv[16];
    const int rg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = rg * 16 + wave * 4 + r;
        float m = 0.f;
        if (row < M) {
            const bf16_t* xr = x + (size_t)row * ldx;
            for (int k = lane * 8; k < K; k += 512) {
                const u32x4 v = *(const u32x4*)(xr + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = f16 ? h2f(v[j] & 0xffff) : bf2f(v[j] & 0xffff), b = f16 ? h2f(v[j] >> 16) : bf2f(v[j] >> 16);
                    m = fmaxf(m, fmaxf(fabsf(a), fabsf(b)));
                }
            }
        }
        m = wave_max(m);
        if (lane == 0) {
            const float sc = m > 0.f ? m * (1.0f / 448.0f) : 1.0f;          // the weights' rule (rowscale_kernel)
            s_inv[wave * 4 + r] = row < M ? 1.0f / sc : 0.f;
            if (row < M && blockIdx.y == 0) xscale[row] = sc;
        }
    }
    __syncthreads();
    const int j = lane & 15, g = lane >> 4;
    const int row = rg * 16 + j;
    const float inv = s_inv[j];
    const bf16_t* xr = x + (size_t)(row < M ? row : 0) * ldx;
    const int KB = K >> 7;
    const int cper = (KB * 2 + gridDim.y - 1) / gridDim.y, c0 = blockIdx.y * cper, c1 = c0 + cper < KB * 2 ? c0 + cper : KB * 2;
    for (int c = c0 + wave; c < c1; c += 4) {            // c = 2q + h: the 16-byte half h of k-block q
        const int k0 = (c >> 1) * 128 + (c & 1) * 64 + g * 8;
        uint32_t o[4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x4 v = *(const u32x4*)(xr + k0 + 32 * s);
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = (f16 ? h2f(v[e] & 0xffff) : bf2f(v[e] & 0xffff)) * inv;
                f[2 * e + 1] = (f16 ? h2f(v[e] >> 16) : bf2f(v[e] >> 16)) * inv;
            }
            int r0 = 0, r1 = 0;
            r0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], r0, false);
            r0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], r0, true);
            r1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], r1, false);
            r1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], r1, true);
            o[2 * s] = (uint32_t)r0; o[2 * s + 1] = (uint32_t)r1;
        }
        xq[((size_t)rg * KB * 2 + c) * 64 + lane] = u32x4{o[0], o[1], o[2], o[3]};
    }
}

// ---------------------------------------------------------------------------------------------- the GEMM
template <int BM>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(LinArgs a, const u32x4* __restrict__ xq, const float* __restrict__ xscale,
                                                          int tiles_m, int tiles_n, int k_per, float* __restrict__ ws) {
    constexpr int MF = BM / 32;                       // 16-row m fragments per wave
    constexpr int XP = BM / 8;                        // 1-KiB X pieces per stage
    constexpr int STAGE = 16384 + XP * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wn = wave >> 1, wm = wave & 1;
    int bid = blockIdx.x;
    {   // XCD-banded tile order (block b runs on XCD b % 8): an XCD works on neighbouring tiles and shares their operands in its L2
        const int nblk = tiles_m * tiles_n, q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid / tiles_m, tile_m = bid - tile_n * tiles_m;      // m fastest: the blocks of a column tile share its weights
    const int KBall = a.K >> 7, KSP = (a.KS + 1) >> 1;
    const int MRG = (a.M + 15) >> 4;
    // split-K (ws != nullptr): blockIdx.y owns k-blocks [q0, q0 + KB) and leaves sx[m] * its partial sums in slab blockIdx.y
    const int q0 = blockIdx.y * k_per;
    const int KB = KBall - q0 < k_per ? KBall - q0 : k_per;

    // staging sources: wave w brings W pieces 4w..4w+3 (row groups 2w, 2w+1; two chunks each) and X pieces (XP/4)w..
    const char* wsrc[4];
    const char* xsrc[XP / 4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int pi = wave * 4 + p;
        int rgg = tile_n * 8 + (pi >> 1);
        if (rgg >= a.NRG) rgg = a.NRG - 1;
        wsrc[p] = (const char*)a.w + ((size_t)rgg * KSP + 2 * q0 + (pi & 1)) * 1024 + lane * 16;
    }
#pragma unroll
    for (int p = 0; p < XP / 4; ++p) {
        const int pj = wave * (XP / 4) + p;
        int rgm = tile_m * (BM / 16) + (pj >> 1);
        if (rgm >= MRG) rgm = MRG - 1;
        xsrc[p] = (const char*)xq + (((size_t)rgm * KBall + q0) * 2 + (pj & 1)) * 1024 + lane * 16;
    }
    auto stage = [&](int q, int slot) {
        char* sb = smem + slot * STAGE;
#pragma unroll
        for (int p = 0; p < 4; ++p) glds16(wsrc[p] + (size_t)q * 2048, sb + (wave * 4 + p) * 1024);
#pragma unroll
        for (int p = 0; p < XP / 4; ++p) glds16(xsrc[p] + (size_t)q * 2048, sb + 16384 + (wave * (XP / 4) + p) * 1024);
    };
    f32x4 acc[4][MF];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = f32x4{0, 0, 0, 0};

    constexpr int NST = 3;
    stage(0, 0);
    if (KB > 1) stage(1, 1);
    int slot = 0;
    for (int q = 0; q < KB; ++q) {
        // this wave's pieces of k-block q have landed (the batch of q + 1 may stay in flight: vmcnt retires in order) ...
        if (q + 1 < KB) {
            if constexpr (XP == 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // raw s_barrier: __syncthreads() is an LDS fence, and with LDS-DMA in flight hipcc turns that into s_waitcnt vmcnt(0) -- the
        // loads issued two k-blocks ahead would be waited for at once (measured: 3 us per k-block, an L2 round trip each)
        __builtin_amdgcn_s_barrier();                             // ... everyone's; and every wave is done reading the slot of k-block q - 1
        if (q + 2 < KB) stage(q + 2, slot == 0 ? 2 : slot - 1);   // = (q + 2) % 3, the slot k-block q - 1 was read from
        const char* sw = smem + slot * STAGE;
        const char* sx = sw + 16384;
        i32x8 bf[MF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int rgm = wm * MF + mf;
            const u32x4 lo = *(const u32x4*)(sx + (rgm * 2) * 1024 + lane * 16), hi = *(const u32x4*)(sx + (rgm * 2 + 1) * 1024 + lane * 16);
            bf[mf] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int rgl = wn * 4 + nf;
            const u32x4 lo = *(const u32x4*)(sw + (rgl * 2) * 1024 + lane * 16), hi = *(const u32x4*)(sw + (rgl * 2 + 1) * 1024 + lane * 16);
            const i32x8 af = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)          // cbsz = blgp = 0: both operands e4m3; block scales e8m0 127 = 1.0
                acc[nf][mf] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf[mf], acc[nf][mf], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
        slot = slot == NST - 1 ? 0 : slot + 1;
    }
    // epilogue: D[n = 4g + j][m = i] of each 16 x 16 block.  Full tiles (the LLM's linears: N % 128 == 0): the row scales, the bias
    // and the residual pieces of a whole row block are fetched up front as 16-byte vectors (store4's per-element scalar loads
    // behind a wait each made the epilogue as long as the k-loop); ragged N / special outputs take the generic store4.
    if (ws) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int m = tile_m * BM + wm * (BM / 2) + mf * 16 + i;
            if (m >= a.M) continue;
            const float sxm = xscale[m];
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const int n0 = tile_n * 128 + wn * 64 + nf * 16 + g * 4;
                if (n0 >= a.N) continue;                            // (split-K needs N % 4 == 0: whole pieces)
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[nf][mf][j] * sxm;
                *(f32x4*)(ws + ((size_t)blockIdx.y * a.M + m) * a.N + n0) = o;
            }
        }
        return;
    }
    const bool fast = (tile_n + 1) * 128 <= a.N && (a.ldo & 3) == 0 && (a.ldo_bf16 & 3) == 0 && (a.ldr & 3) == 0 && a.act == SM_ACT_NONE;
    if (fast) {
        f32x4 sw4[4], b4[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int n0 = tile_n * 128 + wn * 64 + nf * 16 + g * 4;
            sw4[nf] = *(const f32x4*)(a.wscale + n0);
            b4[nf] = a.bias ? *(const f32x4*)(a.bias + n0) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int m = tile_m * BM + wm * (BM / 2) + mf * 16 + i;
            if (m >= a.M) continue;
            const float sxm = xscale[m];
            f32x4 r4[4];
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const int n0 = tile_n * 128 + wn * 64 + nf * 16 + g * 4;
                r4[nf] = a.residual ? *(const f32x4*)(a.residual + (size_t)m * a.ldr + n0) : f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const int n0 = tile_n * 128 + wn * 64 + nf * 16 + g * 4;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[nf][mf][j] * sxm * sw4[nf][j] + b4[nf][j] + r4[nf][j];
                if (a.out_f32) *(f32x4*)(a.out_f32 + (size_t)m * a.ldo + n0) = o;
                if (a.out_bf16) *(u32x2*)(a.out_bf16 + (size_t)m * a.ldo_bf16 + n0) = u32x2{pack16_rt(o[0], o[1], a.f16), pack16_rt(o[2], o[3], a.f16)};
            }
        }
        return;
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int m = tile_m * BM + wm * (BM / 2) + mf * 16 + i;
        const float sxm = m < a.M ? xscale[m] : 0.f;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int n0 = tile_n * 128 + wn * 64 + nf * 16 + g * 4;
            f32x4 v = acc[nf][mf];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= sxm;
            store4(a, m, n0, v, nullptr);
        }
    }
}

// per-HIP-stream workspace of the quantised activations (grown on demand; steady state allocates nothing)
static std::mutex g_xq_mu;
static std::map<hipStream_t, std::pair<void*, size_t>> g_xq;
void release_fp8_workspace(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_xq_mu);
    auto e = g_xq.find(st);
    if (e != g_xq.end()) { if (e->second.first) (void)hipFree(e->second.first); g_xq.erase(e); }
}

int launch_gemm_fp8(LinArgs& a, hipStream_t st) {
    SM_REQUIRE((a.K & 127) == 0 && a.wscale && !a.w2 && !a.vt && a.remap_in == 0, "gemm_fp8: K %% 128 == 0, one fp8 weight image with row scales, plain outputs");
    SM_REQUIRE((a.ldx & 7) == 0, "gemm_fp8: 16-bit x needs ldx %% 8 == 0");
    const int MRG = (a.M + 15) / 16, KB = a.K / 128;
    const size_t qbytes = (size_t)MRG * KB * 2048, need = qbytes + (size_t)MRG * 16 * sizeof(float);
    char* ws;
    {
        std::lock_guard<std::mutex> lk(g_xq_mu);
        auto& e = g_xq[st];
        if (e.second < need) {
            if (e.first) { SM_HIP(hipStreamSynchronize(st)); (void)hipFree(e.first); e.first = nullptr; e.second = 0; }
            SM_HIP(hipMalloc(&e.first, need));
            e.second = need;
        }
        ws = (char*)e.first;
    }
    u32x4* xq = (u32x4*)ws;
    float* xscale = (float*)(ws + qbytes);
    const int qsl = MRG >= 256 ? 1 : (MRG >= 64 ? 4 : 8);                 // column slices per row group: enough blocks to fill the chip
    quant_rows_fp8_kernel<<<dim3(MRG, qsl), 256, 0, st>>>((const bf16_t*)a.x, a.M, a.K, a.ldx, a.f16, xq, xscale);
    SM_LAUNCH_CHECK();
    static bool attr_set = false;
    if (!attr_set) {
        SM_HIP(hipFuncSetAttribute((const void*)gemm_fp8_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (16384 + 16 * 1024)));
        SM_HIP(hipFuncSetAttribute((const void*)gemm_fp8_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (16384 + 8 * 1024)));
        attr_set = true;
    }
    const int tiles_n = cdiv(a.N, 128);
    // 128-row tiles once they fill the chip twice over (two blocks per CU), else 64-row tiles: more blocks, less row padding
    const bool big = (long)cdiv(a.M, 128) * tiles_n >= 512;
    const int tiles_m = cdiv(a.M, big ? 128 : 64);
    // split-K (the kernel can leave sx[m] * partial sums in fp32 slabs for launch_splitk_reduce) was measured at M = 328 on the
    // Mistral shapes and is NOT used: qkv 44 -> 55 us, o 38 -> 46 us, down 113 -> 112 us -- the slab round trip costs what the
    // extra blocks gain (SM_FP8_SPLITK=n forces n slices for experiments).
    int S = 1, k_per = KB;
    {
        static int force = -1;
        if (force < 0) { const char* e = getenv("SM_FP8_SPLITK"); force = e ? atoi(e) : 0; }
        if (force > 1 && !big && (a.N & 3) == 0 && KB >= 2 * force) { k_per = cdiv(KB, force); S = cdiv(KB, k_per); }
    }
    float* slabs = nullptr;
    if (S > 1) { int rc = splitk_workspace(st, (size_t)S * a.M * a.N * sizeof(float), &slabs); if (rc) return rc; }
    const dim3 grid(tiles_m * tiles_n, S);
    if (big) gemm_fp8_kernel<128><<<grid, 256, 3 * (16384 + 16 * 1024), st>>>(a, xq, xscale, tiles_m, tiles_n, k_per, slabs);
    else gemm_fp8_kernel<64><<<grid, 256, 3 * (16384 + 8 * 1024), st>>>(a, xq, xscale, tiles_m, tiles_n, k_per, slabs);
    SM_LAUNCH_CHECK();
    if (S > 1) return launch_splitk_reduce(a, slabs, S, a.N, st);          // fixed slab order: deterministic; wscale / bias / act / residual there
    return SM_OK;
}
