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
// Tile 128(n) x BM(m) x 128(k) per 4-wave block (BM = 128, or 64 when M is small: more blocks, less padding), two LDS stages of
// (16 + BM/8) KiB filled by global_load_lds one k-block ahead, one barrier per k-block; a wave owns 64(n) x BM/2(m).
#include <map>
#include <mutex>

#include "linear_common.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;

// ---------------------------------------------------------------------------------------------- activation quantiser
// one block per 16 rows; wave w finds the row maxima of rows 4w..4w+3, then every thread writes 16-byte fragment pieces.
// x: 16-bit (bf16 or fp16) [M][ldx]; rows >= M quantise to zero.  K % 128 == 0.
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
            if (row < M) xscale[row] = sc;
        }
    }
    __syncthreads();
    const int j = lane & 15, g = lane >> 4;
    const int row = rg * 16 + j;
    const float inv = s_inv[j];
    const bf16_t* xr = x + (size_t)(row < M ? row : 0) * ldx;
    const int KB = K >> 7;
    for (int c = wave; c < KB * 2; c += 4) {             // c = 2q + h: the 16-byte half h of k-block q
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
                                                          int tiles_m, int tiles_n) {
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
    const int KB = a.K >> 7, KSP = (a.KS + 1) >> 1;
    const int MRG = (a.M + 15) >> 4;

    // staging sources: wave w brings W pieces 4w..4w+3 (row groups 2w, 2w+1; two chunks each) and X pieces (XP/4)w..
    const char* wsrc[4];
    const char* xsrc[XP / 4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int pi = wave * 4 + p;
        int rgg = tile_n * 8 + (pi >> 1);
        if (rgg >= a.NRG) rgg = a.NRG - 1;
        wsrc[p] = (const char*)a.w + ((size_t)rgg * KSP + (pi & 1)) * 1024 + lane * 16;
    }
#pragma unroll
    for (int p = 0; p < XP / 4; ++p) {
        const int pj = wave * (XP / 4) + p;
        int rgm = tile_m * (BM / 16) + (pj >> 1);
        if (rgm >= MRG) rgm = MRG - 1;
        xsrc[p] = (const char*)xq + (((size_t)rgm * KB) * 2 + (pj & 1)) * 1024 + lane * 16;
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

    stage(0, 0);
    for (int q = 0; q < KB; ++q) {
        const int slot = q & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of k-block q have landed
        __syncthreads();                                          // ... everyone's; and every wave is done reading the other slot
        if (q + 1 < KB) stage(q + 1, slot ^ 1);
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
    }
    // epilogue: D[n = 4g + j][m = i] of each 16 x 16 block; the activation-row scale here, the weight-row scale, bias, activation,
    // residual and the stores in store4 (linear_common.h)
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
    quant_rows_fp8_kernel<<<MRG, 256, 0, st>>>((const bf16_t*)a.x, a.M, a.K, a.ldx, a.f16, xq, xscale);
    SM_LAUNCH_CHECK();
    static bool attr_set = false;
    if (!attr_set) {
        SM_HIP(hipFuncSetAttribute((const void*)gemm_fp8_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (16384 + 16 * 1024)));
        SM_HIP(hipFuncSetAttribute((const void*)gemm_fp8_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (16384 + 8 * 1024)));
        attr_set = true;
    }
    const int tiles_n = cdiv(a.N, 128);
    // 128-row tiles once they fill the chip twice over (two blocks per CU), else 64-row tiles: more blocks, less row padding
    const bool big = (long)cdiv(a.M, 128) * tiles_n >= 512;
    if (big) {
        const int tiles_m = cdiv(a.M, 128);
        gemm_fp8_kernel<128><<<tiles_m * tiles_n, 256, 2 * (16384 + 16 * 1024), st>>>(a, xq, xscale, tiles_m, tiles_n);
    } else {
        const int tiles_m = cdiv(a.M, 64);
        gemm_fp8_kernel<64><<<tiles_m * tiles_n, 256, 2 * (16384 + 8 * 1024), st>>>(a, xq, xscale, tiles_m, tiles_n);
    }
    SM_LAUNCH_CHECK();
    return SM_OK;
}
