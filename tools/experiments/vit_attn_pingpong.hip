// EXPERIMENT (not built into the library): ViT attention with two wave groups per block one barrier apart ("ping-pong": one
// wave of every SIMD in its matrix phase while the other is in its softmax phase).  Correct (passed tests/test_gpu_ops.py
// test_vit_attention and the 28-frame tests when it was wired into launch_attn), but 119 us against 78 us for vit_attn_kernel at
// B = 28.  Why, measured (tools/experiments/mfma_valu_coexec*.hip): on gfx950 the VALU work of one wave does NOT overlap with the
// MFMAs of another wave of the same SIMD -- 4 waves of 36 MFMAs + 4 waves of 128 v_fma_f32 take the SUM of the two alone (466 ns
// vs 301 / 160), v_pk_fma_f32 / v_pk_mul_f32 even more than the sum; inside one wave 3 fillers per MFMA hide 30 % (v_fma), 36 %
// (v_exp), 68 % (v_max3 / v_cvt_pk) of their cost and packed f32 math costs MORE than it does alone.  So the schedule buys no
// concurrency, and with one 8-wave block per CU (182 VGPRs) the per-block fixed costs (dispatch gap ~2.4 us, q / first-tile
// latency, 13 barriers, the merge: 47.7 us of the 119 with all compute removed) are no longer hidden by co-resident blocks.
// To try again: paste the kernel behind vit_attn_kernel in streammind_amd/csrc/attention.hip and launch it with 512 threads and
// 64 KiB of dynamic LDS from launch_attn's p.v branch.

// ------------------------------------------------------------------------------------------------ ViT, two wave groups
// The same attention with EIGHT waves per block in two groups that split the KEYS (group 0: key tiles 0, 2, 4, ..., group 1: 1, 3,
// 5, ...; wave w and wave w + 4 own the same 32 queries) and run one barrier-delimited interval apart, so that on every SIMD one
// wave is in its matrix phase (PV of tile j-1, then QK^T of tile j: 36 MFMAs) while the other is in its vector phase (softmax of
// its tile: ~130 VALU instructions, and the staging of its next tile).  In vit_attn_kernel all waves of a SIMD run the same
// phases at the same time (the blocks start together and stay in step): SQ_VALU_MFMA_COEXEC was 28 % of the MFMA time with the
// VALU busy 54 % of the kernel -- the two pipes took turns.  Per group and local tile j:
//     M(j): PV(j-1) from slot (j-1)&1, QK^T(j) from slot j&1              V(j): softmax(j); tile j+1 -> registers -> slot (j+1)&1
// group g runs M(0) V(0) M(1) V(1) ... M(n) starting at interval g; every interval ends with one block barrier.  Tiles are staged
// global -> registers -> LDS (4 x 16 B per lane and tile; the LDS image is the one vit_attn_kernel's DMA produces, so the
// fragment reads are the same); slot (j+1)&1 is free when V(j) writes it: its last reader was PV(j-1) in M(j).  The two partial
// softmaxes of a query meet in LDS at the end (flash-decoding merge inside the block).
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <bool F16>
__global__ __launch_bounds__(512, 4) void vit_attn_pp_kernel(AttnP p) {
    constexpr int DH = 64, KROW = 128, TILE_BYTES = 16384;
    extern __shared__ __attribute__((aligned(16))) char lds[];          // [group][slot] tiles of 16 KiB (K 8 KiB | V 8 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wq = wave & 3;
    const int i = lane & 15, g = lane >> 4;
    int h, b, qtile;
    {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int gidx = (j / p.nqt) * 8 + xcd;
        qtile = j % p.nqt;
        if (gidx >= p.H * p.nbatch) return;
        h = gidx % p.H; b = gidx / p.H;
    }
    const int q0 = qtile * 128 + wq * 32;
    const bool active = q0 < p.nq;
    const int nk = p.nk;
    const int NT = (nk + 63) >> 6;
    const int n = (NT - grp + 1) >> 1;                   // tiles of this group
    const int K_total = max(2 * ((NT + 1) >> 1) + 1, 2 * (NT >> 1) + 2);
    char* const mybuf = lds + grp * 2 * TILE_BYTES;

    bf16x8 qf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 16 + i;
        if (qr >= p.nq) qr = p.nq - 1;
        const bf16_t* src = p.q + b * p.q_bs + (long)qr * p.q_rs + h * DH + g * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qb][ks] = *(const bf16x8*)(src + ks * 32);
    }
    f32x4 o[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int df = 0; df < 4; ++df) o[qb][df] = f32x4{0, 0, 0, 0};
    float m_run[2] = {-INFINITY, -INFINITY};
    f32x4 lsum[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    const float one8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    const bf16x8 ones = make8<F16>(one8);

    // per-lane staging sources (the mapping of vit_attn_kernel's DMA pieces, with the group's wave index): wave wq moves K pieces
    // 2wq, 2wq+1 and V pieces 2wq, 2wq+1 (1 KiB = 8 LDS rows each) of the group's current tile
    const bf16_t* ksrc[2];
    const bf16_t* vsrc[2];
    int krow[2], vrow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int R = (wq * 2 + j) * 8 + (lane >> 3), slot = lane & 7;
        const int r5 = R & 31;
        const int key = (R & 32) | (((r5 >> 2) & 3) << 3) | (((r5 >> 4) & 1) << 2) | (r5 & 3);
        krow[j] = key;
        ksrc[j] = p.k + b * p.k_bs + h * DH + (long)(grp * 64 + key) * p.k_rs + ((slot ^ (R & 7)) * 8);
        const int hk = ((R >> 1) & 1) | (((R >> 3) & 1) << 1);
        vrow[j] = R;
        vsrc[j] = p.v + b * p.v_bs + h * DH + (long)(grp * 64 + R) * p.v_rs + ((((slot >> 1) ^ hk) * 2 + (slot & 1)) * 8);
    }
    const long kadv = 128 * p.k_rs, vadv = 128 * p.v_rs;
    u32x4 kreg[2], vreg[2];
    auto fetch = [&](int j) {                             // local tile j -> registers (rows past the last key re-read key nk - 1)
        const int kt0 = (2 * j + grp) * 64;
        const bool clamp = kt0 + 64 > nk;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16_t* ks = ksrc[u];
            const bf16_t* vs = vsrc[u];
            if (clamp) {
                ks -= (long)max(kt0 + krow[u] - (nk - 1), 0) * p.k_rs;
                vs -= (long)max(kt0 + vrow[u] - (nk - 1), 0) * p.v_rs;
            }
            kreg[u] = *(const u32x4*)ks;
            vreg[u] = *(const u32x4*)vs;
            ksrc[u] += kadv;
            vsrc[u] += vadv;
        }
    };
    auto stash = [&](int slot) {                          // registers -> the group's LDS slot
        char* Kl = mybuf + slot * TILE_BYTES;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *(u32x4*)(Kl + (wq * 2 + u) * 1024 + lane * 16) = kreg[u];
            *(u32x4*)(Kl + 8192 + (wq * 2 + u) * 1024 + lane * 16) = vreg[u];
        }
    };

    f32x4 s[2][2][2];
    bf16x8 pf[2][2];
    auto qk = [&](int j) {                                // S^T = K . Q^T of local tile j
        const char* Kl = mybuf + (j & 1) * TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int rho = kb * 32 + f * 16 + i;
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(Kl + rho * KROW + (((ks * 4 + g) ^ (rho & 7)) * 16));
                    a0 = mfma16<F16>(kf, qf[0][ks], a0);
                    a1 = mfma16<F16>(kf, qf[1][ks], a1);
                }
                s[0][kb][f] = a0;
                s[1][kb][f] = a1;
            }
    };
    auto pv = [&](int j) {                                // O^T += V^T . P^T of local tile j (V^T fragments by transpose reads)
        const char* Vl = mybuf + (j & 1) * TILE_BYTES + 8192;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            lsum[0] = mfma16<F16>(ones, pf[0][kb], lsum[0]);
            lsum[1] = mfma16<F16>(ones, pf[1][kb], lsum[1]);
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                union { bf16x8 v; s16x4 hlf[2]; } vf;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int key = kb * 32 + g * 8 + u * 4 + (i >> 2);
                    const int hk = ((key >> 1) & 1) | (((key >> 3) & 1) << 1);
                    vf.hlf[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(Vl + key * 128 + ((df ^ hk) * 32) + (i & 3) * 8));
                }
                o[0][df] = mfma16<F16>(vf.v, pf[0][kb], o[0][df]);
                o[1][df] = mfma16<F16>(vf.v, pf[1][kb], o[1][df]);
            }
        }
    };
    auto softmax = [&](int j) {
        const int kt0 = (2 * j + grp) * 64;
        if (kt0 + 64 > nk) {                              // the cache's last tile: keys past the end count as -inf
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kt0 + kb * 32 + g * 8 + f * 4 + r;
                            s[qb][kb][f][r] = key >= nk ? -INFINITY : s[qb][kb][f][r];
                        }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = vmax3(vmax3(s[qb][0][0][0], s[qb][0][0][1], s[qb][0][0][2]), vmax3(s[qb][0][0][3], s[qb][0][1][0], s[qb][0][1][1]),
                             vmax3(s[qb][0][1][2], s[qb][0][1][3], s[qb][1][0][0]));
            mx = vmax3(mx, vmax3(s[qb][1][0][1], s[qb][1][0][2], s[qb][1][0][3]), vmax3(s[qb][1][1][0], s[qb][1][1][1], s[qb][1][1][2]));
            mx = fmaxf(mx, s[qb][1][1][3]);
            mx = xor32_max(xor16_max(mx));
            const float m_new = fmaxf(m_run[qb], mx);         // finite: every tile holds at least one real key
            const float mc = m_new * p.c;
            const bool moved = m_new != m_run[qb];
            const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run[qb], p.c, -mc));
            m_run[qb] = m_new;
            const f32x2 c2 = {p.c, p.c}, mc2 = {-mc, -mc};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float pe[8];
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 t = f32x2{s[qb][kb][f][r], s[qb][kb][f][r + 1]} * c2 + mc2;
                        pe[f * 4 + r] = __builtin_amdgcn_exp2f(t[0]);
                        pe[f * 4 + r + 1] = __builtin_amdgcn_exp2f(t[1]);
                    }
                pf[qb][kb] = make8<F16>(pe);
            }
            if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
                for (int df = 0; df < 4; ++df) o[qb][df] *= alpha;
                lsum[qb] *= alpha;
            }
        }
    };

    // ---- prologue: every group stages its first tile
    if (n > 0) { fetch(0); stash(0); }
    __syncthreads();
    // ---- intervals
    for (int k = 0; k < K_total; ++k) {
        const int ph = k - grp;
        if (ph >= 0 && ph <= 2 * n && n > 0) {
            const int j = ph >> 1;
            if ((ph & 1) == 0) {                          // matrix phase M(j)
                if (active && !(p.pos0 & 4)) {
                    if (j >= 1) pv(j - 1);
                    if (j < n) qk(j);
                }
            } else {                                      // vector phase V(j)
                const bool more = j + 1 < n && !(p.pos0 & 1);
                if (more) fetch(j + 1);
                if (active && !(p.pos0 & 2)) softmax(j);
                if (more) stash((j + 1) & 1);
            }
        }
        __syncthreads();
    }
    // ---- merge the two key halves of every query: group 1 parks (m, l, o) in LDS, group 0 combines and writes
    float* const park = (float*)lds;                      // [wq][36][64 lanes]
    if (grp == 1 && active) {
        float* dst = park + (size_t)wq * 36 * 64 + lane;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            dst[(qb * 18 + 0) * 64] = m_run[qb];
            dst[(qb * 18 + 1) * 64] = lsum[qb][0];
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(qb * 18 + 2 + df * 4 + r) * 64] = o[qb][df][r];
        }
    }
    __syncthreads();
    if (grp == 0 && active) {
        const float* src = park + (size_t)wq * 36 * 64 + lane;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float mb = src[(qb * 18 + 0) * 64], lb = src[(qb * 18 + 1) * 64];
            const float M = fmaxf(m_run[qb], mb);          // group 0 always holds tile 0: finite
            const float wa = __builtin_amdgcn_exp2f((m_run[qb] - M) * p.c);
            const float wb = mb == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((mb - M) * p.c);
            const float inv = 1.0f / (lsum[qb][0] * wa + lb * wb);
            const int qr = q0 + qb * 16 + i;
            if (qr < p.nq) {
                bf16_t* dst = p.o + b * p.o_bs + (long)qr * p.o_rs + h * DH + g * 4;
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (o[qb][df][r] * wa + src[(qb * 18 + 2 + df * 4 + r) * 64] * wb) * inv;
                    *(u32x2*)(dst + df * 16) = u32x2{pack16<F16>(v[0], v[1]), pack16<F16>(v[2], v[3])};
                }
            }
        }
    }
}

