// Flash-style attention for gfx950, all softmax state lane-local.
//
// Orientation trick (MFMA 16x16x32 bf16, C/D map col = lane & 15, row = (lane >> 4) * 4 + reg):
//   S^T = K . Q^T   : A operand = K rows (from LDS), B operand = Q rows (registers) -> a lane owns ONE query
//                     (col = lane & 15) and 4 keys per fragment, so row max / row sum are in-lane reductions
//                     plus two xor-shuffles (lanes ^16, ^32).
//   O^T = V^T . P^T : A operand = V^T rows (d-major, keys contiguous), B operand = P for the lane's own query.
//   The K rows are PERMUTED when they are written to LDS (row rho = f*16 + i holds key (i>>2)*8 + f*4 + (i&3))
//   so that the S^T accumulators of fragments f = 0,1 are, register for register, the 8 consecutive keys the
//   PV MFMA wants in its B operand: no transpose, no LDS round trip, no cross-lane traffic for P.
//   V is consumed as V^T ([d][keys]); the producers (QKV GEMM epilogue / KV-append kernel) write it that way.
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include "common.h"
#include "host.h"

typedef __attribute__((address_space(3))) void* lds_ptr_a;
typedef const __attribute__((address_space(1))) void* gbl_ptr_a;

struct AttnP {
    const bf16_t* q; long q_bs, q_rs;       // batch stride, row stride (elements); head h at column h*DH
    const bf16_t* k; long k_bs, k_rs;       // kv head kvh at column kvh*DH
    const bf16_t* vt; long vt_bs, vt_hs;    // V^T: [batch][kv head][DH][vt_ld]   (VROW == false)
    int vt_ld;
    const bf16_t* v; long v_bs, v_rs;       // row-major V: [batch][key][...], kv head kvh at column kvh*DH (VROW == true)
    bf16_t* o; long o_bs, o_rs;
    int nq, nk, H, KV, causal, pos0;
    int window;                              // sliding-window attention (Mistral `sliding_window`): query at position q sees keys (q - window, q]; 0 = all
    float c;                                 // softmax scale * log2(e)
    // flash-decoding (single-token decode): blockIdx.z = key split; partial (unnormalised o, m, l) go to part_*
    int split_len;                           // keys per split (multiple of 64), 0 = no splitting
    float* part_o;                           // [splits][H][nq][DH] fp32
    float* part_ml;                          // [splits][H][nq][2]  (running max in scaled-log2 domain source units, l)
    int nqt, nbatch;                         // tiled mode: query tiles per (batch, head), batch count
    int lpt;                                 // causal prefill, one query tile per block: heaviest tiles of ALL heads first, the query heads of one KV group on one XCD (round 6)
    int pair;                                // causal prefill: a block runs query tiles t and nqt - 1 - t one after the other (nqt even): every block walks nqt + 1 key-tile units
                                             // instead of 1 .. nqt (the whole grid is resident at once, so the launch lasted as long as its heaviest tile)
    // batched decode (GROUPQ, blockIdx.x = stream): per-stream caches and key counts; nseg == 0: single stream (k / vt / nk above)
    int nseg;
    long part_bs, ml_bs;                     // per-stream strides of part_o / part_ml (floats)
    SmDecodeSeg seg;
};

template <int DH, bool GROUPQ = false, bool VROW = false, bool CAUSAL = false, bool F16 = false>       // F16: q / K / V^T / P / ctx are IEEE fp16 (llm_fp16)
// GROUPQ: decode mode (query row r of kv-group h is head h*nq + r); VROW: V is row-major [key][d] and is transposed while
// it is staged; CAUSAL is compile-time so that the mask is a handful of v_cmp/v_cndmask under ONE wave-uniform branch (as
// a runtime flag it compiled to two scalar branches per score: 64 per key tile)
__global__ __launch_bounds__(256, 2) void attn_kernel(AttnP p) {
    constexpr int KSQ = DH / 32;             // k-steps of the QK^T contraction
    constexpr int DF = DH / 16;              // d fragments of the output
    constexpr int KROW = DH * 2;             // bytes per K row in LDS
    constexpr int KCH = DH / 8;              // 16-byte chunks per K row
    constexpr int KMASK = KCH - 1 > 15 ? 15 : (KCH - 1);
    constexpr int TILE_BYTES = 64 * KROW + DH * 128;
    __shared__ __attribute__((aligned(16))) char lds[2 * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    // Block -> (query tile, head, batch).  In the tiled (non-split) mode the grid is 1-D and remapped so that all the
    // query tiles of one (batch, head) -- which stream the same K/V -- run on the SAME XCD (block L runs on XCD L % 8):
    // measured without it, each of the 5 q-tiles of a ViT head fetched its K/V through a different L2 (FETCH_SIZE 5x).
    int h, b, qtile0;
    if (GROUPQ) {
        h = blockIdx.y; b = blockIdx.x; qtile0 = 0;         // b: stream index of a batched decode (grid.x == 1 otherwise)
    } else {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int nqt_g = (CAUSAL && p.pair) ? p.nqt >> 1 : p.nqt;
        int grp = (j / nqt_g) * 8 + xcd;
        qtile0 = j % nqt_g;
        if (CAUSAL && p.lpt) {
            // longest first: tile t walks 2 (t + 1) key tiles, so the dispatch order is t = nqt - 1 of every head, then nqt - 2, ... (two blocks per CU, the light
            // tiles fill the tail); XCD x takes heads x G/8 .. (x + 1) G/8 - 1: with grouped-query attention the heads that stream the same K / V share an L2
            const int g8 = (p.H * p.nbatch) >> 3;
            qtile0 = p.nqt - 1 - j / g8;
            grp = xcd * g8 + j % g8;
        }
        if (grp >= p.H * p.nbatch) return;
        h = grp % p.H; b = grp / p.H;
    }
    const int kvh = h / (p.H / p.KV);
    const int npass = (CAUSAL && !GROUPQ && p.pair) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const int qtile = pass ? qtile0 : (npass == 2 ? p.nqt - 1 - qtile0 : qtile0);      // the heavy tile first
    if (pass) __syncthreads();                  // every wave is done with the LDS tiles of the first pass
    const int q0 = qtile * 128 + wave * 32;
    const bool segm = GROUPQ && p.nseg > 0;
    const int nk = segm ? p.seg.pos[b] + 1 : p.nk;

    bf16x8 qf[2][KSQ];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 16 + i;
        if (qr >= p.nq) qr = p.nq - 1;
        const bf16_t* src = GROUPQ ? p.q + b * p.q_bs + ((long)h * p.nq + qr) * DH + g * 8
                                   : p.q + b * p.q_bs + (long)qr * p.q_rs + h * DH + g * 8;
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) qf[qb][ks] = *(const bf16x8*)(src + ks * 32);
    }
    f32x4 o[2][DF];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int df = 0; df < DF; ++df) o[qb][df] = f32x4{0, 0, 0, 0};
    float m_run[2] = {-INFINITY, -INFINITY};
    // row sums ride the matrix pipe: one extra PV fragment whose V^T rows are all ones accumulates sum_k P[q][k] (of the SAME
    // bf16-rounded P the numerator uses) in every row -> no VALU adds per score and no cross-lane reduction at the end
    f32x4 lsum[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    bf16x8 ones;
    {
        const uint32_t one2 = pack16<F16>(1.0f, 1.0f);
        ones = __builtin_bit_cast(bf16x8, u32x4{one2, one2, one2, one2});
    }
    const bool active = q0 < p.nq;

    int k_end = nk, k_begin = 0;
    if (CAUSAL) {
        int last = p.pos0 + min(qtile * 128 + 127, p.nq - 1) + 1;
        k_end = min(k_end, last);
        // sliding window: nothing older than the FIRST query row's window is visible to any row of this tile
        if (p.window > 0) k_begin = max(0, p.pos0 + qtile * 128 - p.window + 1) & ~63;
    }
    // decode (GROUPQ): the one query row is the newest position nk - 1; with a window the key range starts at nk - window
    const int dec_lo = (!CAUSAL && p.window > 0) ? max(0, nk - p.window) : 0;
    if (p.split_len) {
        k_begin = (dec_lo & ~63) + blockIdx.z * p.split_len;
        k_end = min(k_end, k_begin + p.split_len);
    }
    const bf16_t* kbase = (segm ? (const bf16_t*)p.seg.kc[b] : p.k + b * p.k_bs) + kvh * DH;
    const bf16_t* vbase = VROW ? nullptr : (segm ? (const bf16_t*)p.seg.vtc[b] : p.vt + b * p.vt_bs) + kvh * p.vt_hs;

    // K / V^T tiles go HBM/L2 -> registers -> LDS: the loads of tile t+1 are issued before tile t is multiplied (their
    // latency hides under the MFMAs) and written to the OTHER LDS buffer at the top of the next iteration, so there is
    // one __syncthreads per tile.
    constexpr int KJ = (64 * KCH) / 256, VJ = (DH * 8) / 256;
    constexpr int VPJ = (32 * KCH) / 256;          // VROW: (key pair, 16-byte d-chunk) items per thread
    u32x4 kreg[KJ], vreg[VROW ? 2 * VPJ : VJ];
    const bf16_t* vrow_base = VROW ? p.v + b * p.v_bs + kvh * DH : nullptr;
    // per-thread source pointers of tile 0, advanced by one tile per fetch (the per-key clamp is only needed in a tile
    // that runs past nk: recomputing min(key, nk-1) * row_stride for every item cost ~80 VALU instructions per tile)
    const bf16_t* kptr[KJ];
    const bf16_t* vptr[VROW ? 2 * VPJ : VJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        const int c = tid + 256 * j;
        kptr[j] = kbase + (long)(k_begin + c / KCH) * p.k_rs + (c % KCH) * 8;
    }
    if (VROW) {
#pragma unroll
        for (int j = 0; j < VPJ; ++j) {
            const int it = tid + 256 * j;
            vptr[2 * j] = vrow_base + (long)(k_begin + 2 * (it & 31)) * p.v_rs + (it >> 5) * 8;
            vptr[2 * j + 1] = vptr[2 * j] + p.v_rs;
        }
    } else {
#pragma unroll
        for (int j = 0; j < VJ; ++j) {
            const int c = tid + 256 * j;
            vptr[j] = vbase + (long)(c >> 3) * p.vt_ld + k_begin + (c & 7) * 8;
        }
    }
    const long kstep = 64 * p.k_rs, vstep = VROW ? 64 * p.v_rs : 64;
    auto fetch = [&](int kt0) {
        if (kt0 + 64 <= nk || !VROW) {
            if (kt0 + 64 <= nk) {
#pragma unroll
                for (int j = 0; j < KJ; ++j) kreg[j] = *(const u32x4*)kptr[j];
            } else {
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    const int key = (tid + 256 * j) / KCH;
                    kreg[j] = *(const u32x4*)(kptr[j] - (long)max(kt0 + key - (nk - 1), 0) * p.k_rs);
                }
            }
#pragma unroll
            for (int j = 0; j < (VROW ? 2 * VPJ : VJ); ++j) vreg[j] = *(const u32x4*)vptr[j];
        } else {
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const int key = (tid + 256 * j) / KCH;
                kreg[j] = *(const u32x4*)(kptr[j] - (long)max(kt0 + key - (nk - 1), 0) * p.k_rs);
            }
#pragma unroll
            for (int j = 0; j < VPJ; ++j) {
                const int kp = (tid + 256 * j) & 31;
                vreg[2 * j] = *(const u32x4*)(vptr[2 * j] - (long)max(kt0 + 2 * kp - (nk - 1), 0) * p.v_rs);
                vreg[2 * j + 1] = *(const u32x4*)(vptr[2 * j + 1] - (long)max(kt0 + 2 * kp + 1 - (nk - 1), 0) * p.v_rs);
            }
        }
#pragma unroll
        for (int j = 0; j < KJ; ++j) kptr[j] += kstep;
#pragma unroll
        for (int j = 0; j < (VROW ? 2 * VPJ : VJ); ++j) vptr[j] += vstep;
    };
    if (k_begin < k_end) fetch(k_begin);
    int buf = 0;
    for (int kt0 = k_begin; kt0 < k_end; kt0 += 64) {
        char* Kl = lds + buf * TILE_BYTES;
        char* Vl = Kl + 64 * KROW;
        // ---- stage K (64 keys, permuted rows) and V^T (DH rows x 64 keys) tiles
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            int c = tid + 256 * j;
            int key = c / KCH, cc = c % KCH;
            int kk = key & 31;
            int rho = (key & 32) + (((kk >> 2) & 1) << 4) + (((kk >> 3) << 2) | (kk & 3));
            *(u32x4*)(Kl + rho * KROW + ((cc ^ (rho & KMASK)) * 16)) = kreg[j];
        }
        if (VROW) {
#pragma unroll
            for (int j = 0; j < VPJ; ++j) {
                int it = tid + 256 * j;
                int kp = it & 31, cc = it >> 5;
                const u32x4 a = vreg[2 * j], bq = vreg[2 * j + 1];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t lo = (e & 1) ? (a[e >> 1] >> 16) : (a[e >> 1] & 0xffffu);
                    const uint32_t hi = (e & 1) ? (bq[e >> 1] & 0xffff0000u) : (bq[e >> 1] << 16);
                    const int d = cc * 8 + e;
                    *(uint32_t*)(Vl + d * 128 + (((kp >> 2) ^ (d & 7)) * 16) + (kp & 3) * 4) = lo | hi;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < VJ; ++j) {
                int c = tid + 256 * j;
                int d = c >> 3, cc = c & 7;
                *(u32x4*)(Vl + d * 128 + ((cc ^ (d & 7)) * 16)) = vreg[j];
            }
        }
        __syncthreads();
        if (kt0 + 64 < k_end) fetch(kt0 + 64);
        buf ^= 1;

        // a wave whose 32 query rows are all past nq (ViT: 577 = 4 x 128 + 65 -> the last wave of the last q-tile) only helps
        // staging; a tile whose second 32-key block is all past k_end (ViT: key 576 alone in tile 10) skips that block
        if (!active) continue;
        const bool half = kt0 + 32 >= k_end;
        // ---- S^T = K . Q^T  (2 key blocks of 32 x 2 fragments x 2 query blocks)
        f32x4 s[2][2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                if (kb == 1 && half) {
                    s[0][1][f] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    s[1][1][f] = s[0][1][f];
                    continue;
                }
                const int rho = kb * 32 + f * 16 + i;
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KSQ; ++ks) {
                    bf16x8 kf = *(const bf16x8*)(Kl + rho * KROW + (((ks * 4 + g) ^ (rho & KMASK)) * 16));
                    a0 = mfma16<F16>(kf, qf[0][ks], a0);
                    a1 = mfma16<F16>(kf, qf[1][ks], a1);
                }
                s[0][kb][f] = a0;
                s[1][kb][f] = a1;
            }
        // ---- mask + online softmax (lane-local query = lane & 15)
        // (causal: only the tiles that reach past the wave's FIRST query position -- the diagonal ones -- or start before its LAST query's window hold a
        // masked score; below the diagonal the ~130 compare / select instructions per tile are skipped, wave-uniformly)
        const bool diag = CAUSAL && (kt0 + 63 > p.pos0 + q0 || (p.window > 0 && kt0 < p.pos0 + q0 + 32 - p.window));
        if (diag || kt0 + 64 > nk || kt0 < dec_lo) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int qpos = p.pos0 + q0 + qb * 16 + i;
                const int lim = CAUSAL ? min(nk - 1, qpos) : nk - 1;       // last visible key of this lane's query
                const int lo = CAUSAL ? (p.window > 0 ? qpos - p.window + 1 : 0) : dec_lo;      // first visible key
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kt0 + kb * 32 + g * 8 + f * 4 + r;
                            s[qb][kb][f][r] = (key > lim || key < lo) ? -INFINITY : s[qb][kb][f][r];
                        }
            }
        }
        bf16x8 pf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qb][kb][f][r]);
            mx = xor32_max(xor16_max(mx));            // the 4 lanes (g = 0..3) that share this query
            const float m_new = fmaxf(m_run[qb], mx);
            // a fully masked row (causal, tile ahead of the query) keeps m_new = -inf: guard the subtraction
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float mc = m_use * p.c;
            const bool moved = m_new != m_run[qb];                                  // the running max rose in this tile
            const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run[qb], p.c, -mc));   // raw v_exp_f32; -inf -> 0; 1 if !moved
            m_run[qb] = m_new;
            const f32x2 c2 = {p.c, p.c}, mc2 = {-mc, -mc};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                u32x4 pv;
                if (kb == 1 && half) {
                    pf[qb][1] = __builtin_bit_cast(bf16x8, u32x4{0, 0, 0, 0});
                    continue;
                }
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 t = f32x2{s[qb][kb][f][r], s[qb][kb][f][r + 1]} * c2 + mc2;      // v_pk_fma_f32
                        const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                        pv[f * 2 + (r >> 1)] = pack16<F16>(e[0], e[1]);
                    }
                pf[qb][kb] = __builtin_bit_cast(bf16x8, pv);
            }
            // rescale the accumulators only when some lane's max moved (multiplying by alpha == 1 is exact, so skipping it
            // is bit-identical); after the first few key tiles the maxima are stable and the 16 multiplies disappear
            if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
                for (int df = 0; df < DF; ++df) o[qb][df] *= alpha;
                lsum[qb] *= alpha;
            }
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && half) continue;
            lsum[0] = mfma16<F16>(ones, pf[0][kb], lsum[0]);
            lsum[1] = mfma16<F16>(ones, pf[1][kb], lsum[1]);
#pragma unroll
            for (int df = 0; df < DF; ++df) {
                const int d = df * 16 + i;
                bf16x8 vf = *(const bf16x8*)(Vl + d * 128 + (((kb * 4 + g) ^ (d & 7)) * 16));
                o[0][df] = mfma16<F16>(vf, pf[0][kb], o[0][df]);
                o[1][df] = mfma16<F16>(vf, pf[1][kb], o[1][df]);
            }
        }
    }
    // ---- normalise and store: lane (g, q = i) holds d = df*16 + g*4 + 0..3
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l = lsum[qb][0];
        const int qr = q0 + qb * 16 + i;
        if (p.split_len) {
            if (qr < p.nq) {
                const size_t row = ((size_t)blockIdx.z * p.H + h) * p.nq + qr;
                float* dst = p.part_o + (segm ? b * p.part_bs : 0) + row * DH + g * 4;
#pragma unroll
                for (int df = 0; df < DF; ++df) *(f32x4*)(dst + df * 16) = o[qb][df];
                if (g == 0) { float* ml = p.part_ml + (segm ? b * p.ml_bs : 0); ml[row * 2] = m_run[qb]; ml[row * 2 + 1] = l; }
            }
            continue;
        }
        const float inv = 1.0f / l;
        if (qr < p.nq) {
            bf16_t* dst = p.o + b * p.o_bs + (long)qr * p.o_rs + h * DH + g * 4;
#pragma unroll
            for (int df = 0; df < DF; ++df) {
                f32x4 v = o[qb][df] * inv;
                *(u32x2*)(dst + df * 16) = u32x2{pack16<F16>(v[0], v[1]), pack16<F16>(v[2], v[3])};
            }
        }
    }
  }     // pass
}

// ------------------------------------------------------------------------------------------------ LLM causal prefill, head_dim 128 (round 6)
// The causal tile kernel above spends its key tile like this at Mistral-7B's shape (2048 new tokens, 32 heads over 8 K / V heads of 128; rocprofv3 counters,
// profiles/r06_attn_causal_pmc.txt): matrix pipe busy 20-25 % of the launch, VALU 25-30 %, and the ISA shows why -- with K / V staged through 32 + 16 registers
// next to 64 accumulators hipcc has no room to hoist fragment reads, so both products run as `ds_read_b128 x2; s_waitcnt lgkmcnt(0); v_mfma x2`, one exposed
// LDS round trip per pair of MFMAs (32 per tile), and the paired-tile grid (256 blocks of 4 waves) leaves ONE wave per SIMD to hide it.  This kernel keeps the
// arithmetic of attn_kernel<128, false, false, true> instruction for instruction (same fragment order, same softmax: outputs are bit-identical, tested) and changes
// the schedule around it:
//   * K and V^T tiles go global -> LDS by DMA (global_load_lds, 16 B per lane; the K row permutation and both chunk swizzles sit on the per-lane SOURCE
//     address, the LDS image of a DMA being lane-linear): no staging registers, no ds_write, four 64-bit adds of address arithmetic per tile;
//   * the fragments of a 32-key block are read as a BATCH of eight ds_read_b128 into one of two register sets, always one batch ahead of the MFMAs that
//     consume the other set (K block 0, K block 1 | QK 0 | V^T block 0 | QK 1 | V^T block 1 | softmax | PV 0 | PV 1; sched_barriers pin the phases);
//   * one query tile per block, the LONGEST tiles of all heads first (tile t walks 2 (t + 1) key tiles; the light ones fill the tail), two blocks per CU, and the
//     four query heads of a K / V group on one XCD (they stream the same cache rows through the same L2);
//   * the causal compare / select runs only on tiles that reach past the wave's first query (the diagonal), wave-uniformly.
template <bool F16>
__global__ __launch_bounds__(256, 2) void prefill_attn_kernel(AttnP p) {
    constexpr int DH = 128, KROW = 256, TILE_BYTES = 32768, DF = 8, KSQ = 4;
    __shared__ __attribute__((aligned(16))) char lds[2 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    int h, qtile;
    {
        const int L = blockIdx.x, G = p.H;
        if ((G & 7) == 0) { const int xcd = L & 7, j = L >> 3, g8 = G >> 3; qtile = p.nqt - 1 - j / g8; h = xcd * g8 + j % g8; }
        else { qtile = p.nqt - 1 - L / G; h = L % G; }
    }
    const int kvh = h / (p.H / p.KV);
    const int q0 = qtile * 128 + wave * 32;
    const int nk = p.nk;

    bf16x8 qf[2][KSQ];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 16 + i;
        if (qr >= p.nq) qr = p.nq - 1;
        const bf16_t* src = p.q + (long)qr * p.q_rs + h * DH + g * 8;
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) qf[qb][ks] = *(const bf16x8*)(src + ks * 32);
    }
    f32x4 o[2][DF];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int df = 0; df < DF; ++df) o[qb][df] = f32x4{0, 0, 0, 0};
    float m_run[2] = {-INFINITY, -INFINITY};
    f32x4 lsum[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    bf16x8 ones;
    {
        const uint32_t one2 = pack16<F16>(1.0f, 1.0f);
        ones = __builtin_bit_cast(bf16x8, u32x4{one2, one2, one2, one2});
    }
    const bool active = q0 < p.nq;

    int k_end = min(nk, p.pos0 + min(qtile * 128 + 127, p.nq - 1) + 1), k_begin = 0;
    if (p.window > 0) k_begin = max(0, p.pos0 + qtile * 128 - p.window + 1) & ~63;

    // per-lane DMA sources: wave w brings the 1-KiB pieces 4 w .. 4 w + 3 of the K tile (4 permuted key rows of 256 B each) and of the V^T tile (8 dims x 64 keys)
    const bf16_t* ksrc[4];
    const bf16_t* vsrc[4];
    int krow[4];
    {
        const bf16_t* kbase = p.k + kvh * DH;
        const bf16_t* vbase = p.vt + kvh * p.vt_hs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int P = wave * 4 + j;
            const int R = P * 4 + (lane >> 4), slot = lane & 15, r5 = R & 31;
            const int key = (R & 32) | (((r5 >> 2) & 3) << 3) | (((r5 >> 4) & 1) << 2) | (r5 & 3);      // LDS row R holds key inv(R)
            krow[j] = key;
            ksrc[j] = kbase + (long)(k_begin + key) * p.k_rs + ((slot ^ (R & 15)) * 8);
            const int d = P * 8 + (lane >> 3), s8 = lane & 7;
            vsrc[j] = vbase + (long)d * p.vt_ld + k_begin + ((s8 ^ (d & 7)) * 8);
        }
    }
    const long kadv = 64 * p.k_rs;
    auto issue = [&](int kt0, int bufi) {
        char* Kl = lds + bufi * TILE_BYTES;
        char* Vl = Kl + 16384;
        const bool clamp = kt0 + 64 > nk;                 // the tile runs past the cache's last row: keys >= nk re-read row nk - 1 (masked later)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16_t* ks = ksrc[j];
            if (clamp) ks -= (long)max(kt0 + krow[j] - (nk - 1), 0) * p.k_rs;
            __builtin_amdgcn_global_load_lds((gbl_ptr_a)ks, (lds_ptr_a)(Kl + (wave * 4 + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_a)vsrc[j], (lds_ptr_a)(Vl + (wave * 4 + j) * 1024), 16, 0, 0);
            ksrc[j] += kadv;
            vsrc[j] += 64;
        }
    };
    if (k_begin < k_end) issue(k_begin, 0);
    int buf = 0;
    for (int kt0 = k_begin; kt0 < k_end; kt0 += 64) {
        const char* Kl = lds + buf * TILE_BYTES;
        const char* Vl = Kl + 16384;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                  // every wave's pieces of this tile have landed; everyone is done with the other buffer
        if (kt0 + 64 < k_end) issue(kt0 + 64, buf ^ 1);
        buf ^= 1;
        if (!active) continue;
        const bool half = kt0 + 32 >= k_end;          // the tile's second 32-key block lies past the last visible key
        bf16x8 fa[8], fb[8];
        // ---- batch 1, 2: the K fragments of both key blocks
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < KSQ; ++ks) {
                const int rho = f * 16 + i;
                fa[f * 4 + ks] = *(const bf16x8*)(Kl + rho * KROW + (((ks * 4 + g) ^ (rho & 15)) * 16));
            }
        __builtin_amdgcn_sched_barrier(0);        // (keeps the two batches apart: QK 0 then waits for the first eight reads only)
#pragma unroll
        for (int f = 0; f < 2; ++f)               // (read even when the block is skipped: a branch here makes hipcc wait for ALL sixteen reads before the first MFMA)
#pragma unroll
            for (int ks = 0; ks < KSQ; ++ks) {
                const int rho = 32 + f * 16 + i;
                fb[f * 4 + ks] = *(const bf16x8*)(Kl + rho * KROW + (((ks * 4 + g) ^ (rho & 15)) * 16));
            }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 s[2][2][2];
        // ---- S^T = K . Q^T, key block 0
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSQ; ++ks) {
                a0 = mfma16<F16>(fa[f * 4 + ks], qf[0][ks], a0);
                a1 = mfma16<F16>(fa[f * 4 + ks], qf[1][ks], a1);
            }
            s[0][0][f] = a0;
            s[1][0][f] = a1;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- batch 3: the V^T fragments of key block 0 (into the registers QK 0 has just released)
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const int d = df * 16 + i;
            fa[df] = *(const bf16x8*)(Vl + d * 128 + ((g ^ (d & 7)) * 16));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!half) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KSQ; ++ks) {
                    a0 = mfma16<F16>(fb[f * 4 + ks], qf[0][ks], a0);
                    a1 = mfma16<F16>(fb[f * 4 + ks], qf[1][ks], a1);
                }
                s[0][1][f] = a0;
                s[1][1][f] = a1;
            }
        } else {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                s[0][1][f] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                s[1][1][f] = s[0][1][f];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- batch 4: the V^T fragments of key block 1
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const int d = df * 16 + i;
            fb[df] = *(const bf16x8*)(Vl + d * 128 + (((4 + g) ^ (d & 7)) * 16));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- mask (diagonal tiles, the window's edge, the tile past nk) + online softmax (lane-local query = lane & 15)
        const bool diag = kt0 + 63 > p.pos0 + q0 || (p.window > 0 && kt0 < p.pos0 + q0 + 32 - p.window);
        if (diag || kt0 + 64 > nk) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int qpos = p.pos0 + q0 + qb * 16 + i;
                const int lim = min(nk - 1, qpos);                                  // last visible key of this lane's query
                const int lo = p.window > 0 ? qpos - p.window + 1 : 0;              // first visible key
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kt0 + kb * 32 + g * 8 + f * 4 + r;
                            s[qb][kb][f][r] = (key > lim || key < lo) ? -INFINITY : s[qb][kb][f][r];
                        }
            }
        }
        bf16x8 pf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qb][kb][f][r]);
            mx = xor32_max(xor16_max(mx));
            const float m_new = fmaxf(m_run[qb], mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;                 // a fully masked row keeps m_new = -inf: guard the subtraction
            const float mc = m_use * p.c;
            const bool moved = m_new != m_run[qb];
            const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run[qb], p.c, -mc));
            m_run[qb] = m_new;
            const f32x2 c2 = {p.c, p.c}, mc2 = {-mc, -mc};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                u32x4 pv;
                if (kb == 1 && half) {
                    pf[qb][1] = __builtin_bit_cast(bf16x8, u32x4{0, 0, 0, 0});
                    continue;
                }
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 t = f32x2{s[qb][kb][f][r], s[qb][kb][f][r + 1]} * c2 + mc2;
                        const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                        pv[f * 2 + (r >> 1)] = pack16<F16>(e[0], e[1]);
                    }
                pf[qb][kb] = __builtin_bit_cast(bf16x8, pv);
            }
            if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
                for (int df = 0; df < DF; ++df) o[qb][df] *= alpha;
                lsum[qb] *= alpha;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- O^T += V^T . P^T
        lsum[0] = mfma16<F16>(ones, pf[0][0], lsum[0]);
        lsum[1] = mfma16<F16>(ones, pf[1][0], lsum[1]);
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            o[0][df] = mfma16<F16>(fa[df], pf[0][0], o[0][df]);
            o[1][df] = mfma16<F16>(fa[df], pf[1][0], o[1][df]);
        }
        if (!half) {
            lsum[0] = mfma16<F16>(ones, pf[0][1], lsum[0]);
            lsum[1] = mfma16<F16>(ones, pf[1][1], lsum[1]);
#pragma unroll
            for (int df = 0; df < DF; ++df) {
                o[0][df] = mfma16<F16>(fb[df], pf[0][1], o[0][df]);
                o[1][df] = mfma16<F16>(fb[df], pf[1][1], o[1][df]);
            }
        }
    }
    // ---- normalise and store: lane (g, q = i) holds d = df*16 + g*4 + 0..3
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float inv = 1.0f / lsum[qb][0];
        const int qr = q0 + qb * 16 + i;
        if (qr < p.nq) {
            bf16_t* dst = p.o + (long)qr * p.o_rs + h * DH + g * 4;
#pragma unroll
            for (int df = 0; df < DF; ++df) {
                const f32x4 v = o[qb][df] * inv;
                *(u32x2*)(dst + df * 16) = u32x2{pack16<F16>(v[0], v[1]), pack16<F16>(v[2], v[3])};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ ViT fast path
// Non-causal attention with head_dim 64 and V row-major inside the fused qkv buffer (CLIP ViT).  Same math and register
// layout as attn_kernel, but the tiles never pass through registers:
//   * K and V tiles are DMA'd global -> LDS (global_load_lds, 16 B/lane); the K row permutation / chunk swizzle and the V
//     block swizzle are applied on the per-lane SOURCE address (the LDS image of an LDS-DMA is lane-linear);
//   * V stays row-major [key][d] in LDS and the PV A-operand (V^T fragment: 8 keys of one d per lane) is fetched with two
//     ds_read_b64_tr_b16 (hardware 4x4 transpose) -- the register-staged version spent ~26 VALU + 8 ds_write_b32 per thread
//     and tile on the transpose, and measured 26 % of the kernel in staging + its barrier.
//   V LDS image: byte(key, d) = key*128 + (((d >> 4) ^ h(key)) * 32) + (d & 15)*2, h = ((key >> 1) & 1) | (((key >> 3) & 1) << 1):
//   the 8 rows x 32 B that one 32-lane phase of the transpose read touches land on 8 different 32-byte bank groups.
typedef __attribute__((ext_vector_type(4))) short s16x4;
// VIT_TR_ASM=1: the V^T transpose reads as inline asm.  Behind the builtin hipcc waits `vmcnt(0)` in front of the first V^T read of
// a tile (it cannot tell that LDS read from the LDS-DMA of the NEXT tile, issued at the top of this one).  Measured (round 3, same
// box, B = 28): builtin 73.6 us, asm 76.3 us -- the next tile has long landed by then (QK^T + softmax take longer than an L2 round
// trip), and the asm reads pin the schedule.  Kept for A/B, off.
#ifndef VIT_TR_ASM
#define VIT_TR_ASM 0
#endif
template <bool F16>
__device__ __forceinline__ bf16x8 make8(const float (&e)[8]) {
    const u32x4 u = {pack16<F16>(e[0], e[1]), pack16<F16>(e[2], e[3]), pack16<F16>(e[4], e[5]), pack16<F16>(e[6], e[7])};
    return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// cross-lane max steps without the canonicalising v_max_f32 x, x, x pair hipcc puts in front of fmaxf (inputs come from VALU
// permlane swaps, so no XDL hazard applies)
__device__ __forceinline__ float vmax2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float xor16_max_raw(float x) {
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max_raw(float x) {
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
#ifndef VIT_ATTN_WAVES
#define VIT_ATTN_WAVES 4       // waves (of 32 queries) per block: 4 = 128 queries, 8 = 256 queries per staged K / V tile.  Round 3, same box,
                               // B = 28: 8 waves 83.8 us against 75.6 us for 4 -- 40 % less K / V staging per FLOP buys nothing (the
                               // kernel is bound by each wave's own MFMA + VALU stream) and barriers over 8 waves wait longer
#endif
template <bool F16>      // F16: q / k / v / P / ctx are IEEE fp16 (the ViT's optional fp16 mode), else bf16
__global__ __launch_bounds__(VIT_ATTN_WAVES * 64, 8 / VIT_ATTN_WAVES) void vit_attn_kernel(AttnP p) {
    constexpr int NW = VIT_ATTN_WAVES, PPW = 8 / NW;      // 1-KiB K (and V) pieces of a tile each wave brings
    constexpr int DH = 64, KROW = 128, TILE_BYTES = 16384;
    __shared__ __attribute__((aligned(16))) char lds[2 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    int h, b, qtile;
    {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int grp = (j / p.nqt) * 8 + xcd;
        qtile = j % p.nqt;
        if (grp >= p.H * p.nbatch) return;
        h = grp % p.H; b = grp / p.H;
    }
    const int q0 = qtile * (NW * 32) + wave * 32;
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
    const bool active = q0 < p.nq;
    const int nk = p.nk;

    // per-lane DMA sources: wave w issues K pieces 2w, 2w+1 and V pieces 2w, 2w+1 (1 KiB = 8 LDS rows each)
    const bf16_t* ksrc[PPW];
    const bf16_t* vsrc[PPW];
    int krow[PPW], vrow[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int R = (wave * PPW + j) * 8 + (lane >> 3), slot = lane & 7;
        // K: LDS row R holds key inv(R) (rows permuted so that S^T accumulators are PV B-operands), chunk cc at slot cc ^ (R & 7)
        const int r5 = R & 31;
        const int key = (R & 32) | (((r5 >> 2) & 3) << 3) | (((r5 >> 4) & 1) << 2) | (r5 & 3);
        krow[j] = key;
        ksrc[j] = p.k + b * p.k_bs + h * DH + (long)key * p.k_rs + ((slot ^ (R & 7)) * 8);
        // V: LDS row R = key R, 32-byte block (slot >> 1) holds d-block (slot >> 1) ^ h(R)
        const int hk = ((R >> 1) & 1) | (((R >> 3) & 1) << 1);
        vrow[j] = R;
        vsrc[j] = p.v + b * p.v_bs + h * DH + (long)R * p.v_rs + ((((slot >> 1) ^ hk) * 2 + (slot & 1)) * 8);
    }
    // DMA of the tile the source pointers currently point at; CLAMP: the tile runs past nk (keys >= nk re-read row nk-1, they
    // are masked later).  The pointers then advance by one tile -- no per-tile address arithmetic beyond four 64-bit adds.
    const long kadv = 64 * p.k_rs, vadv = 64 * p.v_rs;
    auto issue = [&](int kt0, int bufi, auto clamp) {
        constexpr bool CLAMP = decltype(clamp)::value;
        char* Kl = lds + bufi * TILE_BYTES;
        char* Vl = Kl + 8192;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const bf16_t* ks = ksrc[j];
            const bf16_t* vs = vsrc[j];
            if (CLAMP) {
                ks -= (long)max(kt0 + krow[j] - (nk - 1), 0) * p.k_rs;
                vs -= (long)max(kt0 + vrow[j] - (nk - 1), 0) * p.v_rs;
            }
            __builtin_amdgcn_global_load_lds((gbl_ptr_a)ks, (lds_ptr_a)(Kl + (wave * PPW + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_a)vs, (lds_ptr_a)(Vl + (wave * PPW + j) * 1024), 16, 0, 0);
            ksrc[j] += kadv;
            vsrc[j] += vadv;
        }
    };
    // per-lane part of the V^T transpose-read addresses: key = kb*32 + g*8 + u*4 + (i >> 2) reads row `key` at 32-byte block df ^ h(key), and h(key) = bit 1 | bit 3 << 1 of the
    // key depends on the lane only ((i >> 3) & 1, g & 1) -- four offsets for the whole kernel instead of sixteen address computations per tile (the compiler does not see through the XOR)
    int vlane[4];
    {
        const int hk = ((i >> 3) & 1) | ((g & 1) << 1);
#pragma unroll
        for (int df = 0; df < 4; ++df) vlane[df] = (g * 8 + (i >> 2)) * 128 + ((df ^ hk) * 32) + (i & 3) * 8;
    }
    // one key tile; PART: the (last) tile that runs past nk -- only it carries the key mask and the half-tile skip
    int buf = 0;
    auto tile = [&](int kt0, auto part) {
        constexpr bool PART = decltype(part)::value;
        const char* Kl = lds + buf * TILE_BYTES;
        const char* Vl = Kl + 8192;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();              // every wave's pieces of this tile have landed; everyone is done with the other buffer
        if (!PART) {
            if (kt0 + 128 <= nk) issue(kt0 + 64, buf ^ 1, std::false_type{});
            else if (kt0 + 64 < nk) issue(kt0 + 64, buf ^ 1, std::true_type{});
        }
        buf ^= 1;
        if (!active) return;
        const bool half = PART && kt0 + 32 >= nk;
        // ---- S^T = K . Q^T
        f32x4 s[2][2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                if (PART && kb == 1 && half) {
                    s[0][1][f] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    s[1][1][f] = s[0][1][f];
                    continue;
                }
                const int rho = kb * 32 + f * 16 + i;
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 kf = *(const bf16x8*)(Kl + rho * KROW + (((ks * 4 + g) ^ (rho & 7)) * 16));
                    a0 = mfma16<F16>(kf, qf[0][ks], a0);
                    a1 = mfma16<F16>(kf, qf[1][ks], a1);
                }
                s[0][kb][f] = a0;
                s[1][kb][f] = a1;
            }
        if (PART) {
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
        // vmax3 is inline asm: the hazard recogniser does not know that it reads registers an MFMA has just written (a VALU read of
        // an XDL result needs up to 19 wait states, which hipcc only inserts in front of instructions it can see) -- without this pad
        // the fp16 build returned NaN at 28 frames.  Nothing is scheduled across it.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 pf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // 16 scores per lane -> 8 v_max3_f32 in three independent chains (plain fmaxf compiles to a canonicalising
            // v_max_f32 x, x, x per operand in front of every max: 24 instructions for the same reduction)
            const float mxa = vmax3(vmax3(s[qb][0][0][0], s[qb][0][0][1], s[qb][0][0][2]), s[qb][0][0][3], s[qb][0][1][0]);
            const float mxb = vmax3(vmax3(s[qb][0][1][1], s[qb][0][1][2], s[qb][0][1][3]), s[qb][1][0][0], s[qb][1][0][1]);
            const float mxc = vmax3(vmax3(s[qb][1][0][2], s[qb][1][0][3], s[qb][1][1][0]), s[qb][1][1][1], s[qb][1][1][2]);
            float mx = vmax3(vmax3(mxa, mxb, mxc), s[qb][1][1][3], s[qb][1][1][3]);
            mx = xor32_max_raw(xor16_max_raw(mx));
            const float m_new = vmax2(m_run[qb], mx);         // finite: every tile holds at least one real key
            const float mc = m_new * p.c;
            const bool moved = m_new != m_run[qb];
            const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run[qb], p.c, -mc));
            m_run[qb] = m_new;
            const f32x2 c2 = {p.c, p.c}, mc2 = {-mc, -mc};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float pe[8];
                if (PART && kb == 1 && half) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pe[e] = 0.0f;
                    pf[qb][1] = make8<F16>(pe);
                    continue;
                }
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
        // ---- O^T += V^T . P^T, V^T fragments by transpose reads of the row-major V tile
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (PART && kb == 1 && half) continue;
            lsum[0] = mfma16<F16>(ones, pf[0][kb], lsum[0]);
            lsum[1] = mfma16<F16>(ones, pf[1][kb], lsum[1]);
#if VIT_TR_ASM
            // The transpose reads are inline asm: behind the BUILTIN hipcc put `s_waitcnt vmcnt(0)` in front of the first V^T read of
            // every tile (an LDS access it cannot tell apart from the LDS-DMA of the NEXT tile that was issued at the top of this
            // one) -- the prefetched tile was waited for in the middle of the tile it was meant to hide behind.  Two d-blocks (four
            // reads) per wait; the registers pass through the wait so that the MFMAs cannot be scheduled in front of it.
#pragma unroll
            for (int d2 = 0; d2 < 4; d2 += 2) {
                s16x4 t[2][2];
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int key = kb * 32 + g * 8 + u * 4 + (i >> 2);
                        const int hk = ((key >> 1) & 1) | (((key >> 3) & 1) << 1);
                        const uint32_t ad = (uint32_t)(uintptr_t)(lds_ptr_a)(Vl + key * 128 + (((d2 + dd) ^ hk) * 32) + (i & 3) * 8);
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(t[dd][u]) : "v"(ad));
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1]));
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) {
                    union { bf16x8 v; s16x4 hlf[2]; } vf;
                    vf.hlf[0] = t[dd][0]; vf.hlf[1] = t[dd][1];
                    o[0][d2 + dd] = mfma16<F16>(vf.v, pf[0][kb], o[0][d2 + dd]);
                    o[1][d2 + dd] = mfma16<F16>(vf.v, pf[1][kb], o[1][d2 + dd]);
                }
            }
#else
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                union { bf16x8 v; s16x4 hlf[2]; } vf;
#pragma unroll
                for (int u = 0; u < 2; ++u)       // key = kb*32 + g*8 + u*4 + (i >> 2): the lane's part of the address is vlane[df] (hoisted), kb and u are immediates
                    vf.hlf[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vl + vlane[df] + kb * 4096 + u * 512));
                o[0][df] = mfma16<F16>(vf.v, pf[0][kb], o[0][df]);
                o[1][df] = mfma16<F16>(vf.v, pf[1][kb], o[1][df]);
            }
#endif
        }
    };
    if (nk >= 64) issue(0, 0, std::false_type{}); else issue(0, 0, std::true_type{});
    int kt0 = 0;
    for (; kt0 + 64 <= nk; kt0 += 64) tile(kt0, std::false_type{});
    if (kt0 < nk) tile(kt0, std::true_type{});
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float inv = 1.0f / lsum[qb][0];
        const int qr = q0 + qb * 16 + i;
        if (qr < p.nq) {
            bf16_t* dst = p.o + b * p.o_bs + (long)qr * p.o_rs + h * DH + g * 4;
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                f32x4 v = o[qb][df] * inv;
                *(u32x2*)(dst + df * 16) = u32x2{pack16<F16>(v[0], v[1]), pack16<F16>(v[2], v[3])};
            }
        }
    }
}

// merge the key splits of flash-decoding: out[h][d] = sum_p w_p o_p / sum_p w_p l_p, w_p = exp2((m_p - M) c).
// One block per head; the (m, l) pairs of all splits are read with one load per lane, then each lane owns two d's.
__global__ __launch_bounds__(64) void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                          int splits, int rows, int dh, float c, bf16_t* __restrict__ out,
                                                          long part_bs, long ml_bs, int f16) {
    const int row = blockIdx.x, lane = threadIdx.x;
    part_o += blockIdx.y * part_bs; part_ml += blockIdx.y * ml_bs; out += (size_t)blockIdx.y * rows * dh;     // blockIdx.y: stream of a batched decode
    float m = -INFINITY, l = 0.f;
    if (lane < splits) {
        m = part_ml[((size_t)lane * rows + row) * 2];
        l = part_ml[((size_t)lane * rows + row) * 2 + 1];
    }
    const float M = wave_max(m);
    const float w = (m == -INFINITY) ? 0.f : exp2f((m - M) * c);
    const float den = wave_sum(w * l);
    for (int d = lane; d < dh; d += 64) {
        float num = 0.f;
        // the partial outputs are fetched 8 splits at a time (independent loads in flight together: with one load per loop
        // trip this latency-only kernel paid a memory round trip per split), accumulated in split order
        for (int s0 = 0; s0 < splits; s0 += 8) {
            float po[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) po[u] = s0 + u < splits ? part_o[((size_t)(s0 + u) * rows + row) * dh + d] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < splits) num += __shfl(w, s0 + u, 64) * po[u];
        }
        out[(size_t)row * dh + d] = (bf16_t)cvt16_rt(num / den, f16);
    }
}

// SM_ATTN_PREFILL=0: the causal prefill through attn_kernel (paired tiles / SM_ATTN_PAIR) instead of prefill_attn_kernel (A/B, and the bit-identity test)
static std::atomic<int> g_prefill_kernel{-1};
static bool prefill_kernel_on() {
    int v = g_prefill_kernel.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = getenv("SM_ATTN_PREFILL"); v = e ? atoi(e) : 1; g_prefill_kernel.store(v, std::memory_order_relaxed); }
    return v != 0;
}
extern "C" int sm_set_prefill_attention_kernel(int on) { g_prefill_kernel.store(on < 0 ? -1 : (on ? 1 : 0), std::memory_order_relaxed); return SM_OK; }
static int launch_attn(AttnP& p, int B, int dh, hipStream_t st, bool f16 = false) {
    p.nqt = cdiv(p.nq, 128); p.nbatch = B;
    if (p.v && dh == 64) p.nqt = cdiv(p.nq, VIT_ATTN_WAVES * 32);
    // causal prefill with an even number (>= 4) of query tiles: tiles t and nqt - 1 - t share a block (SM_ATTN_PAIR=0: off, A/B)
    static int pair_on = -1;
    if (pair_on < 0) { const char* e = getenv("SM_ATTN_PAIR"); pair_on = e ? atoi(e) : 1; }
    p.pair = (pair_on == 1 && p.causal && !p.v && !p.split_len && p.nqt >= 4 && (p.nqt & 1) == 0) ? 1 : 0;
    // SM_ATTN_PAIR=2: one tile per block, longest tiles first (see the kernel)
    p.lpt = (pair_on == 2 && p.causal && !p.v && !p.split_len && p.nqt >= 4 && ((p.H * B) & 7) == 0) ? 1 : 0;
    dim3 grid(cdiv(p.H * B, 8) * 8 * (p.pair ? p.nqt >> 1 : p.nqt));
    SmProfScope prof(SM_PROF_ATTN, st);
    SM_REQUIRE(dh == 64 || dh == 128, "attention: head_dim %d not supported (64 or 128)", dh);
    SM_REQUIRE(!(p.v && p.causal), "attention: row-major V is the non-causal (ViT) mode");
    SM_REQUIRE(!f16 || (p.v && dh == 64) || p.causal, "attention: fp16 operands on the ViT fast path (row-major V, head_dim 64) and the causal LLM path");
    if (p.v) {
        if (dh == 64 && f16) vit_attn_kernel<true><<<grid, VIT_ATTN_WAVES * 64, 0, st>>>(p);
        else if (dh == 64) vit_attn_kernel<false><<<grid, VIT_ATTN_WAVES * 64, 0, st>>>(p);
        else attn_kernel<128, false, true><<<grid, 256, 0, st>>>(p);
    } else if (p.causal && dh == 128 && B == 1 && !p.split_len && prefill_kernel_on()) {
        // the prefill kernel (round 6): one 128-query tile per block, longest first
        const dim3 g2(p.H * p.nqt);
        if (f16) prefill_attn_kernel<true><<<g2, 256, 0, st>>>(p);
        else prefill_attn_kernel<false><<<g2, 256, 0, st>>>(p);
    } else if (p.causal) {
        if (f16) {
            if (dh == 64) attn_kernel<64, false, false, true, true><<<grid, 256, 0, st>>>(p);
            else attn_kernel<128, false, false, true, true><<<grid, 256, 0, st>>>(p);
        } else if (dh == 64) attn_kernel<64, false, false, true><<<grid, 256, 0, st>>>(p);
        else attn_kernel<128, false, false, true><<<grid, 256, 0, st>>>(p);
    } else {
        if (dh == 64) attn_kernel<64, false><<<grid, 256, 0, st>>>(p);
        else attn_kernel<128, false><<<grid, 256, 0, st>>>(p);
    }
    SM_LAUNCH_CHECK();
    return SM_OK;
}

extern "C" int sm_vit_attention(const void* qkv, const void* vt, void* ctx, int B, int S, int H, int dh, int vt_ld, int op_dtype,
                                void* stream) {
    SM_REQUIRE(qkv && ctx && B > 0 && S > 0, "sm_vit_attention: bad args");
    SM_REQUIRE(!vt || (vt_ld % 64 == 0 && vt_ld >= cdiv(S, 64) * 64), "sm_vit_attention: vt_ld must be a multiple of 64 covering S");
    AttnP p;
    p.pair = 0; p.lpt = 0;
    p.nseg = 0; p.part_bs = 0; p.ml_bs = 0;
    const long ld = 3L * H * dh;
    p.q = (const bf16_t*)qkv; p.q_bs = (long)S * ld; p.q_rs = ld;
    p.k = (const bf16_t*)qkv + (long)H * dh; p.k_bs = (long)S * ld; p.k_rs = ld;
    p.vt = (const bf16_t*)vt; p.vt_bs = (long)H * dh * vt_ld; p.vt_hs = (long)dh * vt_ld; p.vt_ld = vt_ld;
    p.v = vt ? nullptr : (const bf16_t*)qkv + 2L * H * dh; p.v_bs = (long)S * ld; p.v_rs = ld;   // vt == NULL: V straight from qkv
    p.o = (bf16_t*)ctx; p.o_bs = (long)S * H * dh; p.o_rs = (long)H * dh;
    p.nq = S; p.nk = S; p.H = H; p.KV = H; p.causal = 0; p.pos0 = 0; p.window = 0;
    p.c = (1.0f / sqrtf((float)dh)) * 1.4426950408889634f;
    p.split_len = 0; p.part_o = nullptr; p.part_ml = nullptr;
    return launch_attn(p, B, dh, (hipStream_t)stream, op_dtype == SM_OP_F16);
}

int sm_llm_attention_ex(const void* q, const void* kcache, const void* vtcache, int n, int pos0, int H, int KV,
                        int dh, int S_max, void* ctx, int f16, void* stream, int window) {
    SM_REQUIRE(window >= 0, "sm_llm_attention: window >= 0");
    SM_REQUIRE(q && kcache && vtcache && ctx && n > 0 && pos0 >= 0, "sm_llm_attention: bad args");
    SM_REQUIRE(S_max % 64 == 0 && pos0 + n <= S_max && H % KV == 0, "sm_llm_attention: S_max %% 64, pos0+n <= S_max, H %% KV");
    AttnP p;
    p.pair = 0; p.lpt = 0;
    p.nseg = 0; p.part_bs = 0; p.ml_bs = 0;
    p.q = (const bf16_t*)q; p.q_bs = 0; p.q_rs = (long)H * dh;
    p.k = (const bf16_t*)kcache; p.k_bs = 0; p.k_rs = (long)KV * dh;
    p.vt = (const bf16_t*)vtcache; p.vt_bs = 0; p.vt_hs = (long)dh * S_max; p.vt_ld = S_max;
    p.v = nullptr; p.v_bs = 0; p.v_rs = 0;
    p.o = (bf16_t*)ctx; p.o_bs = 0; p.o_rs = (long)H * dh;
    p.nq = n; p.nk = pos0 + n; p.H = H; p.KV = KV; p.causal = 1; p.pos0 = pos0; p.window = window;
    p.c = (1.0f / sqrtf((float)dh)) * 1.4426950408889634f;
    p.split_len = 0; p.part_o = nullptr; p.part_ml = nullptr;
    return launch_attn(p, 1, dh, (hipStream_t)stream, f16 != 0);
}
extern "C" int sm_llm_attention(const void* q, const void* kcache, const void* vtcache, int n, int pos0, int H, int KV,
                                int dh, int S_max, void* ctx, void* stream) {
    return sm_llm_attention_ex(q, kcache, vtcache, n, pos0, H, KV, dh, S_max, ctx, 0, stream, 0);
}
extern "C" int sm_llm_attention_window(const void* q, const void* kcache, const void* vtcache, int n, int pos0, int H, int KV,
                                       int dh, int S_max, int window, void* ctx, void* stream) {
    return sm_llm_attention_ex(q, kcache, vtcache, n, pos0, H, KV, dh, S_max, ctx, 0, stream, window);
}


// ------------------------------------------------------------------------------------------------ decode, one launch
// Single-token decode attention with the key-split merge INSIDE the block (no partials in HBM, no merge kernel): one 8-wave
// block per (stream, KV group); wave w owns the 32-key blocks w, w + 8, ... of the cache and every fragment it needs is a
// direct 16-byte global load in MFMA operand layout (nothing is staged through LDS):
//   S^T = K . Q^T : A = K rows (lane (i, g) reads dims ks*32 + g*8.. of key key0 + (i>>2)*8 + a*4 + (i&3): the row permutation
//                   that makes the two accumulators a = 0, 1 the PV B operand register for register), B = the group's query heads;
//   O^T = V^T . P^T: A = V^T rows (lane (i, g) reads positions key0 + g*8.. of dim df*16 + i: contiguous in the d-major cache).
// All K and V loads of a round (two key blocks per wave = 512 keys per block) are issued before anything is consumed: one memory
// round trip per round.  The 8 per-wave partial softmaxes meet in LDS (fp32, 16 KiB) and 512 threads write the merged bf16
// context.  Same arithmetic as attn_kernel + attn_combine_kernel (scores in the scaled log2 domain, P rounded to bf16 before
// both the PV product and the row sum); used while the context is short enough for one CU per KV group (the launcher decides).
template <class SEG>
struct DecAttnPT {
    const bf16_t* q; bf16_t* ctx;      // [S][H*128]
    const bf16_t* k; const bf16_t* vt; // single stream (nseg == 0)
    int nk, H, KV, S_max, nseg;
    int window;                        // 0 = all keys; else the newest `window` keys (the query is the newest position)
    float c;
    SEG seg;
};
typedef DecAttnPT<SmDecodeSeg> DecAttnP;
// NW waves per block (8, or 4 for the batched step: round 6).  The kernel needs 213 registers, i.e. two waves per SIMD: an 8-wave block owns its CU, so its phases
// -- q, one K / V round trip to HBM, softmax + PV, the merge through LDS -- run strictly one after the other per CU (11.5 us per block at 336 keys, 3.8 TB/s over a
// 512-stream step).  Two 4-wave blocks per CU at the same registers overlap one block's round trip with the other's arithmetic and merge.
template <bool F16, class P = DecAttnP, int NW = 8>
__global__ __launch_bounds__(NW * 64) void decode_attn_kernel(P p) {
    constexpr int DH = 128;
    __shared__ float osh[NW][16][DH + 4];
    __shared__ float msh[NW][16], lsh[NW][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int b = blockIdx.x, kvh = blockIdx.y;
    const int rep = p.H / p.KV;
    const bool segm = p.nseg > 0;
    const int nk = segm ? seg_pos(p.seg, b) + 1 : p.nk;
    const bf16_t* kc = (segm ? (const bf16_t*)p.seg.kc[b] : p.k) + kvh * DH;
    const bf16_t* vt = (segm ? (const bf16_t*)p.seg.vtc[b] : p.vt) + (size_t)kvh * DH * p.S_max;
    const long k_rs = (long)p.KV * DH;
    bf16x8 qf[4];
    {
        union { bf16x8 v; u32x4 u; } z;
        z.u = u32x4{0, 0, 0, 0};
        const bf16_t* src = p.q + ((size_t)b * p.H + kvh * rep + (i < rep ? i : 0)) * DH + g * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = i < rep ? *(const bf16x8*)(src + ks * 32) : z.v;
    }
    f32x4 o[8];
#pragma unroll
    for (int df = 0; df < 8; ++df) o[df] = f32x4{0, 0, 0, 0};
    float m_run = -INFINITY, l_run = 0.f;
    const int NB = (nk + 31) >> 5;
    const int k_lo = p.window > 0 ? max(0, nk - p.window) : 0;          // first visible key
    for (int blk0 = (k_lo >> 5) + wave; blk0 < NB; blk0 += 2 * NW) {
        const bool two = blk0 + NW < NB;
        bf16x8 kf[2][2][4], vf[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const int key0 = (blk0 + NW * u) * 32;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                int key = key0 + (i >> 2) * 8 + a * 4 + (i & 3);
                key = key < nk ? key : nk - 1;
                const bf16_t* src = kc + (long)key * k_rs + g * 8;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[u][a][ks] = *(const bf16x8*)(src + ks * 32);
            }
#pragma unroll
            for (int df = 0; df < 8; ++df) vf[u][df] = *(const bf16x8*)(vt + (size_t)(df * 16 + i) * p.S_max + key0 + g * 8);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const int key0 = (blk0 + NW * u) * 32;
            f32x4 sc[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = mfma16<F16>(kf[u][a][ks], qf[ks], acc);
                sc[a] = acc;
            }
            const bool part = key0 + 32 > nk || key0 < k_lo;        // only the last block of the cache (and the first of a window) carries masked keys
            if (part) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + g * 8 + a * 4 + r >= nk || key0 + g * 8 + a * 4 + r < k_lo) sc[a][r] = -INFINITY;
                // positions past the cache end may hold anything (a caller-owned cache need not be zeroed): 0 * garbage must stay 0
#pragma unroll
                for (int df = 0; df < 8; ++df) {
                    union { bf16x8 v; uint16_t h[8]; } t;
                    t.v = vf[u][df];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (key0 + g * 8 + e >= nk) t.h[e] = 0;
                    vf[u][df] = t.v;
                }
            }
            float mx = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])), fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
            mx = xor32_max(xor16_max(mx));
            const float m_new = fmaxf(m_run, mx);            // finite: the block holds at least one real key
            const float mc = m_new * p.c;
            const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run, p.c, -mc));
            m_run = m_new;
            float pe[8], psum = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(sc[a][r], p.c, -mc));
                    const float er = up16<F16>(cvt16<F16>(e));            // the row sum adds the SAME rounded P the PV product multiplies
                    pe[a * 4 + r] = er;
                    psum += er;
                }
            const bf16x8 pf = make8<F16>(pe);
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int df = 0; df < 8; ++df) {
                o[df] *= alpha;
                o[df] = mfma16<F16>(vf[u][df], pf, o[df]);
            }
        }
    }
    // per-wave partial -> LDS (lane (i, g): query i, dims df*16 + g*4 + r; l still split over the four g's)
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (i < rep) {
#pragma unroll
        for (int df = 0; df < 8; ++df) *(f32x4*)&osh[wave][i][df * 16 + g * 4] = o[df];
        if (g == 0) { msh[wave][i] = m_run; lsh[wave][i] = l_run; }
    }
    __syncthreads();
    for (int idx = tid; idx < rep * DH; idx += NW * 64) {
        const int qi = idx >> 7, d = idx & (DH - 1);
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, msh[w][qi]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float mw = msh[w][qi];
            const float wt = mw == -INFINITY ? 0.f : exp2f((mw - M) * p.c);
            num += wt * osh[w][qi][d];
            den += wt * lsh[w][qi];
        }
        p.ctx[((size_t)b * p.H + kvh * rep + qi) * DH + d] = (bf16_t)cvt16<F16>(num / den);
    }
}
// SM_DECODE_ATTN_FUSED=0: keep the split + merge launch pair for every context length (A/B switch).  The one-launch kernel gives
// ONE CU per (stream, KV group): measured at Mistral-7B shapes it equals the launch pair at 328 keys for a single stream and
// loses beyond ~512 (1024 keys: 301 vs 322 tokens/s) -- but with 16+ streams its S x KV blocks fill the chip by themselves and
// it wins (32 streams at 328 keys: 7026 vs 6674 tokens/s aggregate).
// waves per block of the one-launch kernel in a BATCHED step (SM_DECODE_ATTN_NW = 8 | 4; see the kernel)
static int decode_attn_group_waves() {
    static int nw = -1;
    // same box, interleaved twice (profiles/r06_decode_attn_nw_ab.txt), ms per batched step with 8 / 4 waves: 64 streams 4.84 / 4.79, 128: 6.29 / 6.18, 256: 10.19 / 9.98,
    // 512: 16.54 / 16.09 (+1.2 .. +2.8 % tokens/s); 32 streams equal (the small pack keeps 8 waves below 512 blocks)
    if (nw < 0) { const char* e = getenv("SM_DECODE_ATTN_NW"); nw = e ? atoi(e) : 4; if (nw != 4) nw = 8; }
    return nw;
}
static bool decode_attn_fused_ok(int nk, int dh, int S, int KV) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SM_DECODE_ATTN_FUSED"); on = e ? atoi(e) : 1; }
    if (!on || dh != 128) return false;
    return nk <= 384 || (S * KV >= 128 && nk <= 2048);
}
// Single-token decode ("flash-decoding"): the H/KV query heads of one KV group play the role of the query rows of the
// tile kernel (so K/V of a group are streamed once for all its heads), the keys are split across gridDim.z blocks so
// that every CU streams part of the cache, and a tiny kernel merges the partial softmaxes.
// workspace: fp32 [splits_max * H * (dh + 2)]
int sm_llm_decode_attention_ex(const void* q, const void* kcache, const void* vtcache, int pos, int H, int KV, int dh,
                               int S_max, float* workspace, int splits_max, void* ctx, int f16, void* stream, int window) {
    SM_REQUIRE(window >= 0, "sm_llm_decode_attention: window >= 0");
    SM_REQUIRE(q && kcache && vtcache && ctx && workspace && pos >= 0 && pos < S_max, "sm_llm_decode_attention: bad args");
    SM_REQUIRE(S_max % 64 == 0 && H % KV == 0 && H / KV <= 16 && splits_max >= 1 && splits_max <= 64 && dh <= 128, "sm_llm_decode_attention: dims");
    const int rep = H / KV, nk = pos + 1;
    const int k_lo64 = window > 0 ? (nk - window > 0 ? (nk - window) & ~63 : 0) : 0;      // first key tile that holds a visible key
    const int nk_eff = nk - k_lo64;                                                      // keys the kernels walk
    if (decode_attn_fused_ok(nk_eff, dh, 1, KV)) {
        DecAttnP d;
        d.q = (const bf16_t*)q; d.ctx = (bf16_t*)ctx; d.k = (const bf16_t*)kcache; d.vt = (const bf16_t*)vtcache;
        d.nk = nk; d.H = H; d.KV = KV; d.S_max = S_max; d.nseg = 0; d.window = window; d.c = (1.0f / sqrtf((float)dh)) * 1.4426950408889634f;
        SmProfScope prof(SM_PROF_ATTN, (hipStream_t)stream);
        if (f16) decode_attn_kernel<true><<<dim3(1, KV), 512, 0, (hipStream_t)stream>>>(d);
        else decode_attn_kernel<false><<<dim3(1, KV), 512, 0, (hipStream_t)stream>>>(d);
        SM_LAUNCH_CHECK();
        return SM_OK;
    }
    int splits = cdiv(nk_eff, 64);            // one 64-key tile per block while the cache is short: the kernel is a latency chain per tile
    if (splits > splits_max) splits = splits_max;
    const int split_len = cdiv(cdiv(nk_eff, splits), 64) * 64;
    splits = cdiv(nk_eff, split_len);
    AttnP p;
    p.pair = 0; p.lpt = 0;
    p.nseg = 0; p.part_bs = 0; p.ml_bs = 0; p.window = window;
    p.q = (const bf16_t*)q; p.q_bs = 0; p.q_rs = dh;                 // "query row" r of group h <-> head h*rep + r
    p.k = (const bf16_t*)kcache; p.k_bs = 0; p.k_rs = (long)KV * dh;
    p.vt = (const bf16_t*)vtcache; p.vt_bs = 0; p.vt_hs = (long)dh * S_max; p.vt_ld = S_max;
    p.v = nullptr; p.v_bs = 0; p.v_rs = 0;
    p.o = nullptr; p.o_bs = 0; p.o_rs = 0;
    p.nq = rep; p.nk = nk; p.H = KV; p.KV = KV; p.causal = 0; p.pos0 = 0;
    p.c = (1.0f / sqrtf((float)dh)) * 1.4426950408889634f;
    p.split_len = split_len;
    p.part_o = workspace; p.part_ml = workspace + (size_t)splits_max * H * dh;
    p.nqt = 1; p.nbatch = 1;
    // head h of the tile kernel reads q columns h*DH: make group h start at head h*rep by scaling the row stride trick:
    // q pointer for group h = q + h*rep*dh  ->  handled through q_bs = 0 and a per-group base below (grid.y = KV)
    hipStream_t st = (hipStream_t)stream;
    {
        SmProfScope prof(SM_PROF_ATTN, st);
        dim3 grid(1, KV, splits);
        if (dh != 64 && dh != 128) SM_FAIL(SM_EINVAL, "attention: head_dim %d not supported (64 or 128)", dh);
        if (f16) {
            if (dh == 64) attn_kernel<64, true, false, false, true><<<grid, 256, 0, st>>>(p);
            else attn_kernel<128, true, false, false, true><<<grid, 256, 0, st>>>(p);
        } else if (dh == 64) attn_kernel<64, true><<<grid, 256, 0, st>>>(p);
        else attn_kernel<128, true><<<grid, 256, 0, st>>>(p);
        SM_LAUNCH_CHECK();
    }
    attn_combine_kernel<<<H, 64, 0, st>>>(p.part_o, p.part_ml, splits, H, dh, p.c, (bf16_t*)ctx, 0, 0, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
extern "C" int sm_llm_decode_attention(const void* q, const void* kcache, const void* vtcache, int pos, int H, int KV, int dh,
                                       int S_max, float* workspace, int splits_max, void* ctx, void* stream) {
    return sm_llm_decode_attention_ex(q, kcache, vtcache, pos, H, KV, dh, S_max, workspace, splits_max, ctx, 0, stream, 0);
}
extern "C" int sm_llm_decode_attention_window(const void* q, const void* kcache, const void* vtcache, int pos, int H, int KV, int dh,
                                              int S_max, int window, float* workspace, int splits_max, void* ctx, void* stream) {
    return sm_llm_decode_attention_ex(q, kcache, vtcache, pos, H, KV, dh, S_max, workspace, splits_max, ctx, 0, stream, window);
}

// up to SM_BIG_SEG streams in ONE launch of the one-launch kernel (per-stream pointers: 2.5 KB of kernel arguments).  Returns 1 (not
// an error) when the longest context is beyond what that kernel is used for: the caller then goes through sm_llm_decode_attention_seg in
// chunks of SM_MAX_SEG.
int sm_llm_decode_attention_seg_big(const void* q, const SmDecodeSegBig& seg, int S, int H, int KV, int dh, int S_max, void* ctx, int f16, void* stream, int window) {
    SM_REQUIRE(q && ctx && S > 0 && S <= SM_BIG_SEG && S_max % 64 == 0 && H % KV == 0 && H / KV <= 16, "sm_llm_decode_attention_seg_big: bad args");
    int nk = 1;
    for (int t = 0; t < S; ++t) {
        SM_REQUIRE(seg.pos[t] >= 0 && seg.pos[t] < S_max && seg.kc[t] && seg.vtc[t], "sm_llm_decode_attention_seg_big: stream %d: bad position / cache", t);
        nk = seg.pos[t] + 1 > nk ? seg.pos[t] + 1 : nk;
    }
    const int nk_eff = window > 0 && nk > window + 63 ? window + 63 : nk;
    if (!decode_attn_fused_ok(nk_eff, dh, S, KV)) return 1;
    DecAttnPT<SmDecodeSegBig> d;
    d.q = (const bf16_t*)q; d.ctx = (bf16_t*)ctx; d.k = nullptr; d.vt = nullptr;
    d.nk = nk; d.H = H; d.KV = KV; d.S_max = S_max; d.nseg = S; d.seg = seg; d.window = window; d.c = (1.0f / sqrtf((float)dh)) * 1.4426950408889634f;
    SmProfScope prof(SM_PROF_ATTN, (hipStream_t)stream);
    if (decode_attn_group_waves() == 4) {
        if (f16) decode_attn_kernel<true, DecAttnPT<SmDecodeSegBig>, 4><<<dim3(S, KV), 256, 0, (hipStream_t)stream>>>(d);
        else decode_attn_kernel<false, DecAttnPT<SmDecodeSegBig>, 4><<<dim3(S, KV), 256, 0, (hipStream_t)stream>>>(d);
    } else if (f16) decode_attn_kernel<true, DecAttnPT<SmDecodeSegBig>><<<dim3(S, KV), 512, 0, (hipStream_t)stream>>>(d);
    else decode_attn_kernel<false, DecAttnPT<SmDecodeSegBig>><<<dim3(S, KV), 512, 0, (hipStream_t)stream>>>(d);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// single-token decode attention of S streams in ONE launch pair: stream t's query row block q[t] (H heads) against ITS cache
// [0, pos[t]]; the key range is cut into the same number of splits for every stream (sized for the longest context; a split
// that lies beyond a shorter stream's cache contributes weight 0 to the merge)
int sm_llm_decode_attention_seg_tab(const void* q, const SmDecodeSegTab& tab, int S, int nk_max, int H, int KV, int dh, int S_max, void* ctx, int f16, void* stream, int window) {
    SM_REQUIRE(q && ctx && S > 0 && S_max % 64 == 0 && H % KV == 0 && H / KV <= 16 && tab.kc && tab.vtc && tab.pos0 && nk_max >= 1 && nk_max <= S_max,
               "sm_llm_decode_attention_seg_tab: bad args");
    const int nk_eff = window > 0 && nk_max > window + 63 ? window + 63 : nk_max;
    if (!decode_attn_fused_ok(nk_eff, dh, S, KV)) return 1;
    DecAttnPT<SmDecodeSegTab> d;
    d.q = (const bf16_t*)q; d.ctx = (bf16_t*)ctx; d.k = nullptr; d.vt = nullptr;
    d.nk = nk_max; d.H = H; d.KV = KV; d.S_max = S_max; d.nseg = S; d.seg = tab; d.window = window; d.c = (1.0f / sqrtf((float)dh)) * 1.4426950408889634f;
    SmProfScope prof(SM_PROF_ATTN, (hipStream_t)stream);
    if (decode_attn_group_waves() == 4) {
        if (f16) decode_attn_kernel<true, DecAttnPT<SmDecodeSegTab>, 4><<<dim3(S, KV), 256, 0, (hipStream_t)stream>>>(d);
        else decode_attn_kernel<false, DecAttnPT<SmDecodeSegTab>, 4><<<dim3(S, KV), 256, 0, (hipStream_t)stream>>>(d);
    } else if (f16) decode_attn_kernel<true, DecAttnPT<SmDecodeSegTab>><<<dim3(S, KV), 512, 0, (hipStream_t)stream>>>(d);
    else decode_attn_kernel<false, DecAttnPT<SmDecodeSegTab>><<<dim3(S, KV), 512, 0, (hipStream_t)stream>>>(d);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
int sm_llm_decode_attention_seg(const void* q, const SmDecodeSeg& seg, int S, int H, int KV, int dh, int S_max, float* workspace,
                                int splits_max, void* ctx, int f16, void* stream, int window) {
    SM_REQUIRE(q && ctx && workspace && S > 0 && S <= SM_MAX_SEG, "sm_llm_decode_attention_seg: bad args");
    SM_REQUIRE(S_max % 64 == 0 && H % KV == 0 && H / KV <= 16 && splits_max >= 1 && splits_max <= 64 && (dh == 64 || dh == 128), "sm_llm_decode_attention_seg: dims");
    const int rep = H / KV;
    int nk = 1;
    for (int t = 0; t < S; ++t) {
        SM_REQUIRE(seg.pos[t] >= 0 && seg.pos[t] < S_max && seg.kc[t] && seg.vtc[t], "sm_llm_decode_attention_seg: stream %d: bad position / cache", t);
        nk = seg.pos[t] + 1 > nk ? seg.pos[t] + 1 : nk;
    }
    // with a window every stream walks at most window + 63 keys (its own tile-aligned start is computed in the kernel)
    const int nk_eff = window > 0 && nk > window + 63 ? window + 63 : nk;
    if (decode_attn_fused_ok(nk_eff, dh, S, KV)) {
        DecAttnP d;
        d.q = (const bf16_t*)q; d.ctx = (bf16_t*)ctx; d.k = nullptr; d.vt = nullptr;
        d.nk = nk; d.H = H; d.KV = KV; d.S_max = S_max; d.nseg = S; d.seg = seg; d.window = window; d.c = (1.0f / sqrtf((float)dh)) * 1.4426950408889634f;
        SmProfScope prof(SM_PROF_ATTN, (hipStream_t)stream);
        if (decode_attn_group_waves() == 4 && S * KV >= 512) {
            if (f16) decode_attn_kernel<true, DecAttnP, 4><<<dim3(S, KV), 256, 0, (hipStream_t)stream>>>(d);
            else decode_attn_kernel<false, DecAttnP, 4><<<dim3(S, KV), 256, 0, (hipStream_t)stream>>>(d);
        } else if (f16) decode_attn_kernel<true><<<dim3(S, KV), 512, 0, (hipStream_t)stream>>>(d);
        else decode_attn_kernel<false><<<dim3(S, KV), 512, 0, (hipStream_t)stream>>>(d);
        SM_LAUNCH_CHECK();
        return SM_OK;
    }
    int splits = cdiv(nk_eff, 64);
    if (splits > splits_max) splits = splits_max;
    const int split_len = cdiv(cdiv(nk_eff, splits), 64) * 64;
    splits = cdiv(nk_eff, split_len);
    AttnP p;
    p.pair = 0; p.lpt = 0;
    p.window = window;
    p.q = (const bf16_t*)q; p.q_bs = (long)H * dh; p.q_rs = dh;
    p.k = nullptr; p.k_bs = 0; p.k_rs = (long)KV * dh;
    p.vt = nullptr; p.vt_bs = 0; p.vt_hs = (long)dh * S_max; p.vt_ld = S_max;
    p.v = nullptr; p.v_bs = 0; p.v_rs = 0;
    p.o = nullptr; p.o_bs = 0; p.o_rs = 0;
    p.nq = rep; p.nk = nk; p.H = KV; p.KV = KV; p.causal = 0; p.pos0 = 0;
    p.c = (1.0f / sqrtf((float)dh)) * 1.4426950408889634f;
    p.split_len = split_len;
    p.nseg = S; p.seg = seg;
    p.part_bs = (long)splits_max * H * (dh + 2); p.ml_bs = p.part_bs;
    p.part_o = workspace; p.part_ml = workspace + (size_t)splits_max * H * dh;
    p.nqt = 1; p.nbatch = 1;
    hipStream_t st = (hipStream_t)stream;
    {
        SmProfScope prof(SM_PROF_ATTN, st);
        dim3 grid(S, KV, splits);
        if (f16) {
            if (dh == 64) attn_kernel<64, true, false, false, true><<<grid, 256, 0, st>>>(p);
            else attn_kernel<128, true, false, false, true><<<grid, 256, 0, st>>>(p);
        } else if (dh == 64) attn_kernel<64, true><<<grid, 256, 0, st>>>(p);
        else attn_kernel<128, true><<<grid, 256, 0, st>>>(p);
        SM_LAUNCH_CHECK();
    }
    attn_combine_kernel<<<dim3(H, S), 64, 0, st>>>(p.part_o, p.part_ml, splits, H, dh, p.c, (bf16_t*)ctx, p.part_bs, p.ml_bs, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
