// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/streammind_hip.h"

extern thread_local char g_sm_err[512];

#define SM_FAIL(code, ...)                                   \
    do {                                                     \
        snprintf(g_sm_err, sizeof(g_sm_err), __VA_ARGS__);   \
        return (code);                                       \
    } while (0)

#define SM_REQUIRE(cond, ...)                                \
    do {                                                     \
        if (!(cond)) SM_FAIL(SM_EINVAL, __VA_ARGS__);        \
    } while (0)

#define SM_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) SM_FAIL(SM_EHIP, "%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

#define SM_LAUNCH_CHECK() SM_HIP(hipGetLastError())

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Per-segment (= per-stream) base pointers handed to kernels BY VALUE: a connector pass covers at most SM_MAX_SEG streams
// (one weight pass = at most 32 rows), so no device-side pointer table has to be built or uploaded per call.
#define SM_MAX_SEG 32
#define SM_GROUP_DECODE_MAX 512     // active streams of one batched decode step (beyond SM_MAX_SEG: tiled GEMMs over all rows + per-stream kernels in packs)
int sm_skinny_lds64_on();           // linear.hip: products of 33..64 16-bit rows take the LDS-shared weight-streaming kernel (dual weights allowed there)
#define SM_BIG_SEG 128              // streams whose per-stream pointers travel in ONE by-value pack (2.5 KB of kernel arguments)
struct SmSegStates { float* p[SM_MAX_SEG]; };
int sm_mamba_conv_step_seg(const float* xz, int S, int F, int di, int d_conv, const SmSegStates& st, const float* conv_w,
                           const float* conv_b, float* xc, void* stream);                                    // vecops.hip
int sm_mamba_ssm_step_seg(const float* xc, const float* delta, const float* x_dbl, int ldx, int dt_rank, const float* xz, int S,
                          int F, int di, int d_state, const float* A_log, const float* Dp, const SmSegStates& st, float* y,
                          void* stream);
int sm_scatter_rows(const float* src, int S, int F, int d, const SmSegStates& dst, void* stream);
int sm_gate_tail(const float* x, int M, int D, int ldx, const float* gamma, float eps, const float* head_w_f32, float* logits, int32_t* decisions, void* stream);   // vecops.hip
int sm_pool_patches16(const void* h, int B, int S, int D, float* out, int f16, void* stream);          // vecops.hip: patch mean of a 16-bit matrix
int sm_unpack_rows_f32(const void* wp, int N, int K, int f16, float* out, void* stream);

// batched single-token decode over S streams (one row per stream): per-stream KV caches and positions, by value
template <int N> struct SmDecodeSegT { void* kc[N]; void* vtc[N]; int pos[N]; };
// The same per-stream pointers as a TABLE in device memory (round 6): any number of streams in ONE launch.  kc / vtc point at this layer's row of the table
// the batched decode call uploaded once (the pointers do not change during a call: the caches were brought to their common capacity before it), pos0 at the
// streams' positions when the call started; `step` = decode steps since.
struct SmDecodeSegTab { void* const* kc; void* const* vtc; const int* pos0; int step; };
template <class SEG> static __device__ __forceinline__ int seg_pos(const SEG& s, int b) { return s.pos[b]; }
template <> __device__ __forceinline__ int seg_pos<SmDecodeSegTab>(const SmDecodeSegTab& s, int b) { return s.pos0[b] + s.step; }
typedef SmDecodeSegT<SM_MAX_SEG> SmDecodeSeg;
typedef SmDecodeSegT<SM_BIG_SEG> SmDecodeSegBig;      // every stream of a large batched decode step in ONE launch (2.5 KB of kernel arguments)
template <int N> struct SmTokPtrsT { int32_t* p[N]; };
typedef SmTokPtrsT<SM_MAX_SEG> SmTokPtrs;
typedef SmTokPtrsT<SM_BIG_SEG> SmTokPtrsBig;             // the token words of up to SM_BIG_SEG streams in one launch (1 KB of kernel arguments)
// epilogue of the decode-step q/k/v product with RoPE + KV append fused in (linear.hip sm_linear_qkv_rope): row i of the
// activations is stream i's token at seg.pos[i]; q goes out rotated as bf16 [M][H*dh], k rotated into seg.kc[i], v transposed
// into seg.vtc[i] -- the arithmetic of rope_kv_kernel on the fp32 accumulators, without the fp32 round trip and its launch
struct SmRopeEpi { const float* cos_tab; const float* sin_tab; void* q; int H, KV, S_max; SmDecodeSeg seg; };
int sm_linear_qkv_rope(const sm_linear_t* p, const SmRopeEpi& re, void* stream);                                  // linear.hip
int sm_rope_kv_append_seg(const float* qkv, int S, int H, int KV, int dh, const float* cos_tab, const float* sin_tab, void* q_bf16,
                          const SmDecodeSeg& seg, int S_max, int f16, void* stream);                              // vecops.hip
// fp16-aware forms of C-ABI glue ops (f16 != 0: the 16-bit tables / outputs are IEEE fp16, llm_fp16 mode); the extern "C" names keep bf16
int sm_rope_kv_append_ex(const float* qkv, int n, int pos0, int H, int KV, int dh, const float* cos_tab, const float* sin_tab, void* q,
                         void* kcache, void* vtcache, int S_max, int f16, void* stream);
int sm_embed_splice_ex(const int32_t* ids, int n, const void* table, const float* tokens, int D, float* out, int f16, int vocab,
                       int n_tok, void* stream);
int sm_swiglu_ex(const float* gu, int M, int F, void* out, int f16, void* stream);
// window: Mistral's sliding_window (a query at position p sees keys (p - window, p]); 0 = full causal
int sm_llm_decode_attention_seg(const void* q_bf16, const SmDecodeSeg& seg, int S, int H, int KV, int dh, int S_max, float* workspace,
                                int splits_max, void* ctx_bf16, int f16, void* stream, int window = 0);           // attention.hip
// the one-launch decode attention / RoPE + KV append for up to SM_BIG_SEG streams; the attention returns SM_EINVAL-free `1` when the
// contexts are too long for the one-launch kernel (the caller then falls back to chunks of SM_MAX_SEG through the calls above)
int sm_llm_decode_attention_seg_big(const void* q_bf16, const SmDecodeSegBig& seg, int S, int H, int KV, int dh, int S_max, void* ctx_bf16, int f16,
                                    void* stream, int window = 0);
int sm_rope_kv_append_seg_big(const float* qkv, int S, int H, int KV, int dh, const float* cos_tab, const float* sin_tab, void* q_bf16,
                              const SmDecodeSegBig& seg, int S_max, int f16, void* stream);                       // vecops.hip
// ... and for any number of streams through the device-side table (nk_max = the longest context incl. the new token, the caller's own bookkeeping);
// the attention returns 1 like the _big form when the contexts are too long for the one-launch kernel
int sm_llm_decode_attention_seg_tab(const void* q_bf16, const SmDecodeSegTab& tab, int S, int nk_max, int H, int KV, int dh, int S_max, void* ctx_bf16, int f16,
                                    void* stream, int window);                                                       // attention.hip
int sm_rope_kv_append_seg_tab(const float* qkv, int S, int H, int KV, int dh, const float* cos_tab, const float* sin_tab, void* q_bf16,
                              const SmDecodeSegTab& tab, int S_max, int f16, void* stream, int nslab = 0, size_t slab_stride = 0);   // vecops.hip (nslab > 0: qkv = unsummed split-K slabs)
int sm_llm_attention_ex(const void* q, const void* kcache, const void* vtcache, int n, int pos0, int H, int KV, int dh, int S_max,
                        void* ctx, int f16, void* stream, int window = 0);
int sm_llm_decode_attention_ex(const void* q, const void* kcache, const void* vtcache, int pos, int H, int KV, int dh, int S_max,
                               float* workspace, int splits_max, void* ctx, int f16, void* stream, int window = 0);
int sm_embed_tokens_seg(const SmTokPtrs& tok, int S, const void* table_bf16, int D, float* out, const SmTokPtrs& out_rows, int col,
                        int f16, void* stream);
int sm_argmax_rows_seg(const float* logits, int S, int V, int ld, const SmTokPtrs& out, void* stream);
// linear.hip: sm_linear that may leave its split-K slabs UNSUMMED for the next kernel to sum on load (round 6: the batched decode's q|k|v product at 33..128
// streams is 5 K slabs + a reduce launch + the RoPE launch that reads the sums -- the RoPE kernel sums the slabs itself).  Only a plain product (no bias,
// activation, residual, 16-bit output) on the 128 x 128 split-K path leaves slabs: out->S = their count (raw fp32 [S][M][N], stride M * N floats, valid until the
// next tiled product on this HIP stream); out->S == 0: the call wrote p->out_f32 as sm_linear would.
struct SmSlabOut { const float* ws; int S; size_t stride; };
int sm_linear_leave_slabs(const sm_linear_t* p, SmSlabOut* out, void* stream);
// linear.hip: grow the per-HIP-stream split-K slabs / unfused-SwiGLU rows to at least these sizes NOW (set-up time), so that no request allocates
int sm_linear_reserve(hipStream_t st, size_t slab_bytes, size_t dual_bytes);
// the same two per-stream kernels of a batched decode step for up to SM_BIG_SEG streams per launch (round 6: at 512 streams the token gather and the arg-max were
// 16 launches of 32 rows each, 0.39 ms of a 16.4 ms step)
int sm_embed_tokens_seg_big(const SmTokPtrsBig& tok, int S, const void* table_bf16, int D, float* out, const SmTokPtrsBig& out_rows, int col, int f16, void* stream);
int sm_argmax_rows_seg_big(const float* logits, int S, int V, int ld, const SmTokPtrsBig& out, void* stream);

// Optional in-library kernel timing (bench.py's roofline leg): when a class bit is enabled, every launch of that
// class is bracketed by HIP events recorded ON THE LAUNCH STREAM; sm_prof_read() synchronises and sums.
enum { SM_PROF_GEMM = 0, SM_PROF_SKINNY = 1, SM_PROF_ATTN = 2, SM_PROF_NCLS = 3 };
void sm_prof_begin_(int cls, hipStream_t st, long long tag);
void sm_prof_end_(int cls, hipStream_t st);
extern int g_sm_prof_mask;
struct SmProfScope {
    int cls; hipStream_t st; bool on;
    SmProfScope(int c, hipStream_t s, long long tag = 0) : cls(c), st(s), on((g_sm_prof_mask >> c) & 1) { if (on) sm_prof_begin_(cls, st, tag); }      // tag: sm_prof_read_tag's key
    ~SmProfScope() { if (on) sm_prof_end_(cls, st); }
};
