// Path-level orchestration behind the C ABI: weight registry (HF/reference names -> packed device storage),
// workspaces, the ViT forward, the per-frame connector+gate step and the Mistral prefill / greedy decode.
// Everything here is host code issuing the kernels of linear.hip / attention.hip / vecops.hip on one stream;
// there is no host synchronisation on any hot call.
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "host.h"

int sm_pack_weight_ks(const void* w, int N, int K, int ldw, int KS, void* out, void* stream);   // linear.hip
void release_stream_workspaces(hipStream_t st);                                                 // linear.hip
extern "C" size_t sm_packed_fp8_bytes(int N, int K);
extern "C" int sm_quant_pack_weight_fp8(const void* w, int N, int K, int ldw, void* out, float* scale_out, void* stream);

// ------------------------------------------------------------------------------------------------ small kernels
__global__ void bf16_to_f32_kernel(const bf16_t* in, float* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bf2f(in[i]);
}
__global__ void f32_to_bf16_kernel(const float* in, bf16_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (bf16_t)f2bf(in[i]);
}
__global__ void f16_to_bf16_kernel(const _Float16* in, bf16_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (bf16_t)f2bf((float)in[i]);
}
__global__ void bf16_to_f16_kernel(const bf16_t* in, bf16_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (bf16_t)f2h(bf2f(in[i]));          // exact inside fp16's normal range (10 > 7 mantissa bits)
}
__global__ void f32_to_f16_kernel(const float* in, bf16_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (bf16_t)f2h(in[i]);
}
__global__ void f16_to_f32_kernel(const _Float16* in, float* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
__global__ void embed_last_kernel(const int32_t* tok, const bf16_t* table, int D, float* out, int f16) {
    int id = *tok;
    for (int c = threadIdx.x; c < D; c += blockDim.x) out[c] = f16 ? h2f(table[(size_t)id * D + c]) : bf2f(table[(size_t)id * D + c]);
}
__global__ void copy_i32_kernel(const int32_t* src, int32_t* dst) { *dst = *src; }
// the tail of a decode step in ONE launch: greedy argmax of the logits (lowest index wins ties, as torch.argmax) -> pending token;
// emitted as the next step's output id and its embedding row gathered for the next step's first kernel (both optional).  Was
// argmax + copy + gather: three latency-only launches per token.
__global__ __launch_bounds__(1024) void argmax_emit_embed_kernel(const float* __restrict__ lg, int V, int32_t* __restrict__ next_tok,
                                                                 int32_t* __restrict__ emit, const bf16_t* __restrict__ table, int D,
                                                                 float* __restrict__ emb, int f16) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ int winner;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const float t = lg[v];
        if (t > best) { best = t; idx = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { bv[w] = best; bi[w] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
            if (bv[k] > best || (bv[k] == best && bi[k] < idx)) { best = bv[k]; idx = bi[k]; }
        *next_tok = idx;
        if (emit) *emit = idx;
        winner = idx;
    }
    __syncthreads();
    if (emb) {
        const int id = winner;
        for (int c = threadIdx.x; c < D; c += blockDim.x) emb[c] = f16 ? h2f(table[(size_t)id * D + c]) : bf2f(table[(size_t)id * D + c]);
    }
}

// ------------------------------------------------------------------------------------------------ storage
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t b, bool zero = false) {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = b;
        if (b == 0) return SM_OK;
        SM_HIP(hipMalloc(&p, b));
        if (zero) SM_HIP(hipMemset(p, 0, b));
        return SM_OK;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;                 // owns its allocation: movable (a grown K / V cache replaces the old buffers), never copied
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { if (p) (void)hipFree(p); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    template <typename T> T* as() const { return (T*)p; }
};

struct Slot {            // one tensor the path reads
    DevBuf buf;
    int kind = 0;        // 0 = fp32 vector/matrix (as is), 1 = packed linear (possibly a fused group), 2 = bf16 row-major
    int N = 0, K = 0;    // logical dims of the (fused) matrix
    int parts = 1;
    uint32_t loaded = 0; // bit i: part i has been handed over (a reload overwrites, it does not count twice)
    bool complete() const { return loaded == (parts >= 32 ? 0xffffffffu : (1u << parts) - 1u); }
    int Klogical = 0;    // > 0: checkpoint K (the packed image pads it to K); patch embedding only
    bool fp8 = false;    // packed as fp8 + per-row scales (weights_fp8 mode, gate + LLM linears)
    bool wo = false;     // fp8 slot that is weight-only in EVERY fp8 mode (the gate: its rows are never quantised, so a pass of 17..32 rows streams the fp8 image)
    bool f16 = false;    // packed image holds IEEE fp16 instead of bf16 (vit_fp16 mode, ViT linears)
    DevBuf scale;        // fp32 [N]
};

struct sm_model {
    sm_config_t c;
    std::unordered_map<std::string, Slot> slots;           // canonical name -> storage
    struct Route { std::string slot; int row0; int rows; int part; };   // checkpoint name -> (slot, first row, row count, index of this part)
    std::unordered_map<std::string, Route> routes;
    std::vector<std::string> ignored_prefixes;
    bool finalized = false;
    // ViT workspaces, one set per HIP stream the tower is driven on (two streams of one model may then run the tower
    // concurrently on different HIP streams: their kernels fill each other's launch gaps and tails, +7 % aggregate frames/s
    // measured with two 28-frame streams); allocated on the first call of a stream, never afterwards
    struct VitWs { DevBuf patches, x, xn, qkv, ctx, hmid, hbar, stats; };          // stats: LayerNorm-fold partial row sums [rows][D / 256][2] fp32
    std::map<void*, std::unique_ptr<VitWs>> vit_ws;
    std::mutex ws_mu;
    int S = 0, P = 0, Spad = 0, Kpe = 0, Bmax = 0;
    int vit_workspace(void* stream, VitWs** out) {
        std::lock_guard<std::mutex> lk(ws_mu);
        auto& e = vit_ws[stream];
        if (!e) {
            std::unique_ptr<VitWs> w(new VitWs());
            const size_t rows = (size_t)Bmax * S;
            const int D = c.vit_hidden;
            int rc;
            if ((rc = w->patches.alloc((size_t)Bmax * P * Kpe * 2))) return rc;
            if ((rc = w->x.alloc(rows * D * 4))) return rc;
            if ((rc = w->xn.alloc(rows * D * 2))) return rc;
            if ((rc = w->qkv.alloc(rows * 3 * D * 2))) return rc;
            if ((rc = w->ctx.alloc(rows * D * 2))) return rc;
            if ((rc = w->hmid.alloc(rows * c.vit_mlp * 2))) return rc;
            if ((rc = w->hbar.alloc((size_t)Bmax * c.vit_mlp * 4))) return rc;
            if ((rc = w->stats.alloc(rows * (size_t)cdiv(D, 256) * 2 * 4))) return rc;
            e = std::move(w);
        }
        *out = e.get();
        return SM_OK;
    }
    // LLM activation workspaces, one set per HIP stream the decoder is driven on (round 6; they were members of every sm_stream: 193 MB each at
    // Mistral-7B widths, i.e. 97 GB of idle scratch beside 512 open streams).  A call's rows live here only for the duration of the call -- what a stream
    // carries from call to call (pending token, last logits, K / V, frame tokens, Mamba state) stays in the stream -- so streams driven on one HIP stream
    // share a set in stream order, and a reply on the LLM lane (its own HIP stream) has its own.
    struct LlmWs { DevBuf emb, xnb, qkvf, qb, ctxb, actb, attn_ws; };
    static constexpr int LLM_CHUNK = 2048;          // rows of a prefill / teacher-forced chunk
    std::map<void*, std::unique_ptr<LlmWs>> llm_ws;
    int llm_workspace(void* stream, LlmWs** out) {
        std::lock_guard<std::mutex> lk(ws_mu);
        auto& e = llm_ws[stream];
        if (!e) {
            std::unique_ptr<LlmWs> w(new LlmWs());
            const size_t ch = LLM_CHUNK;
            const int ld = c.llm_hidden, dh = ld / c.llm_heads, kn = c.llm_kv_heads * dh, qn = c.llm_heads * dh;
            int rc;
            if ((rc = w->emb.alloc(ch * ld * 4))) return rc;
            if ((rc = w->xnb.alloc(ch * ld * 2))) return rc;
            if ((rc = w->qkvf.alloc(ch * (qn + 2 * kn) * 4))) return rc;
            if ((rc = w->qb.alloc(ch * qn * 2))) return rc;
            if ((rc = w->ctxb.alloc(ch * qn * 2))) return rc;
            if ((rc = w->actb.alloc(ch * c.llm_mlp * 2))) return rc;
            if ((rc = w->attn_ws.alloc((size_t)32 * c.llm_heads * (dh + 2) * 4))) return rc;       // SM_DECODE_SPLITS (= 32) partial softmaxes per head
            // the products' own scratch on this HIP stream, worst case of a chunk, reserved now (never grown inside a request): up to 4 K slabs of
            // [chunk][hidden] fp32 (o_proj / down_proj of a 1024..2048-row chunk) and the [rows][2 mlp] fp32 gate | up rows of a SwiGLU product that
            // does not run fused (<= 256 rows outside the weight-streaming kernel; SM_SWIGLU_FUSE=0 grows it on demand)
            if ((rc = sm_linear_reserve((hipStream_t)stream, (size_t)4 * ch * ld * 4, (size_t)256 * 2 * c.llm_mlp * 4))) return rc;
            e = std::move(w);
        }
        *out = e.get();
        return SM_OK;
    }
    // second tower lane of a caller stream (sm_vit_encode with >= SM_VIT_LANE_MIN frames): a side HIP stream + fork / join events
    // `more`: further side streams (+ their join events) of the frame lanes of a SMALL call (2..8 frames: one lane per frame, see vit_small_lanes)
    struct Lane { hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; std::vector<hipStream_t> more; std::vector<hipEvent_t> more_join; };
    std::map<void*, Lane> lanes;
    int lane_of(void* stream, Lane** out, int n_more = 0) {
        std::lock_guard<std::mutex> lk(ws_mu);
        Lane& L = lanes[stream];
        if (!L.side) {
            SM_HIP(hipStreamCreateWithFlags(&L.side, hipStreamNonBlocking));
            SM_HIP(hipEventCreateWithFlags(&L.fork, hipEventDisableTiming));
            SM_HIP(hipEventCreateWithFlags(&L.join, hipEventDisableTiming));
        }
        while ((int)L.more.size() < n_more) {
            hipStream_t st; hipEvent_t ev;
            SM_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            L.more.push_back(st);
            SM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            L.more_join.push_back(ev);
        }
        *out = &L;
        return SM_OK;
    }
    ~sm_model() {
        for (auto& kv : lanes) {
            if (kv.second.side) { (void)hipStreamSynchronize(kv.second.side); release_stream_workspaces(kv.second.side); (void)hipStreamDestroy(kv.second.side); }
            if (kv.second.fork) (void)hipEventDestroy(kv.second.fork);
            if (kv.second.join) (void)hipEventDestroy(kv.second.join);
            for (auto st : kv.second.more) { (void)hipStreamSynchronize(st); release_stream_workspaces(st); (void)hipStreamDestroy(st); }
            for (auto ev : kv.second.more_join) (void)hipEventDestroy(ev);
        }
    }
    DevBuf gate_head_f32;          // the gate's 2-way head as fp32 [2][d] (16-bit checkpoint values widened): operand of the fused gate tail
    // RoPE tables for the LLM
    DevBuf rope_cos, rope_sin;
    int rope_len = 0;
    // every tensor the hot calls touch, resolved ONCE at finalize (no string building / hashing per launch; element
    // addresses of an unordered_map are stable)
    struct LayerW { const Slot *qkv = nullptr, *out = nullptr, *fc1 = nullptr, *fc2 = nullptr, *v = nullptr, *o = nullptr, *gu = nullptr, *down = nullptr;
                    const float *qkv_b = nullptr, *out_b = nullptr, *fc1_b = nullptr, *fc2_b = nullptr, *ln1_w = nullptr, *ln1_b = nullptr,
                                *ln2_w = nullptr, *ln2_b = nullptr;
                    // ViT LayerNorm folding (sm_linear_t.fold_*): W . gamma and W . beta + b of the LayerNorm in front of q|k|v and of fc1 (fp32 [N], made at finalize)
                    const float *qkv_g = nullptr, *qkv_c = nullptr, *fc1_g = nullptr, *fc1_c = nullptr; };
    std::vector<DevBuf> fold_vecs;
    struct Resolved {
        const Slot *patch = nullptr, *pre = nullptr, *in_proj = nullptr, *x_proj = nullptr, *dt_proj = nullptr, *out_proj = nullptr, *post = nullptr,
                   *gate_head = nullptr, *lm_head = nullptr, *embed = nullptr;
        const float *cls = nullptr, *pos = nullptr, *pre_ln_w = nullptr, *pre_ln_b = nullptr;
        const float *pre_b = nullptr, *cn_w = nullptr, *cn_b = nullptr, *conv_w = nullptr, *conv_b = nullptr, *dt_b = nullptr, *A_log = nullptr, *Dp = nullptr,
                    *nf_w = nullptr, *nf_b = nullptr, *post_b = nullptr, *gate_norm = nullptr, *llm_norm = nullptr;
        std::vector<LayerW> vit, gate, llm;
    } R;

    Slot* get(const std::string& n) {
        auto it = slots.find(n);
        return it == slots.end() ? nullptr : &it->second;
    }
    template <typename T> T* ptr(const std::string& n) { return slots.at(n).buf.as<T>(); }
};

static void add_linear(sm_model* m, const std::string& slot, int N, int K, const std::vector<std::string>& names, int rows_each) {
    Slot& s = m->slots[slot];
    s.kind = 1; s.N = N; s.K = K; s.parts = (int)names.size();
    for (size_t i = 0; i < names.size(); ++i) m->routes[names[i]] = {slot, (int)i * rows_each, rows_each, (int)i};
}
static void add_f32(sm_model* m, const std::string& name, int N, int K = 1) {
    Slot& s = m->slots[name];
    s.kind = 0; s.N = N; s.K = K;
    m->routes[name] = {name, 0, N, 0};
}
static void add_f32_fused(sm_model* m, const std::string& slot, int N, const std::vector<std::string>& names, int rows_each) {
    Slot& s = m->slots[slot];
    s.kind = 0; s.N = N; s.K = 1; s.parts = (int)names.size();
    for (size_t i = 0; i < names.size(); ++i) m->routes[names[i]] = {slot, (int)i * rows_each, rows_each, (int)i};
}

extern "C" int sm_model_create(const sm_config_t* cfg, sm_model** out) {
    SM_REQUIRE(cfg && out, "sm_model_create: null arg");
    const sm_config_t& c = *cfg;
    SM_REQUIRE(c.vit_hidden % 64 == 0 && c.vit_mlp % 64 == 0 && c.vit_hidden % c.vit_heads == 0, "vit dims must be multiples of 64");
    SM_REQUIRE(c.vit_hidden / c.vit_heads == 64 || c.vit_hidden / c.vit_heads == 128, "vit head_dim must be 64 or 128");
    SM_REQUIRE(c.vit_image % c.vit_patch == 0, "image size must be a multiple of the patch size");
    SM_REQUIRE(c.conn_mm_hidden == c.vit_hidden, "connector input width must equal the ViT width");
    SM_REQUIRE(c.conn_d_model % 32 == 0 && c.conn_d_state >= 0 && c.conn_d_state <= 32 && c.conn_d_conv <= 8, "connector dims");
    // conn_d_state == 0: no Mamba connector and no event gate in this model (stock VideoLLaMA2 checkpoints: their projector -- the
    // STC family -- runs above the C ABI, streammind_amd/model/stc_connector.py, and hands its tokens over with
    // sm_stream_write_tokens); the tower and the LLM are unchanged, the push entry points refuse
    const bool has_conn = c.conn_d_state > 0;
    SM_REQUIRE(has_conn || c.gate_layers == 0, "a model without connector (conn_d_state == 0) has no gate: gate_layers must be 0");
    SM_REQUIRE(c.gate_hidden == c.conn_d_model, "gate width must equal the connector width");
    SM_REQUIRE(c.gate_hidden % 32 == 0 && c.gate_mlp % 32 == 0 && c.gate_heads % c.gate_kv_heads == 0, "gate dims");
    SM_REQUIRE(c.max_frames_per_call >= 1, "max_frames_per_call >= 1");
    SM_REQUIRE(!c.vit_fp16 || c.vit_hidden / c.vit_heads == 64, "vit_fp16 needs head_dim 64 (the ViT attention fast path)");
    if (c.llm_layers > 0) {
        SM_REQUIRE(c.llm_hidden == c.conn_d_model, "LLM width must equal the connector width");
        SM_REQUIRE(c.llm_hidden % 64 == 0 && c.llm_mlp % 64 == 0, "LLM dims must be multiples of 64");
        SM_REQUIRE(c.llm_hidden / c.llm_heads == 64 || c.llm_hidden / c.llm_heads == 128, "LLM head_dim must be 64 or 128");
    }
    sm_model* m = new sm_model();
    m->c = c;
    const int D = c.vit_hidden, g = c.vit_image / c.vit_patch;
    m->P = g * g; m->S = m->P + 1; m->Spad = cdiv(m->S, 64) * 64;
    m->Kpe = cdiv(3 * c.vit_patch * c.vit_patch, 64) * 64;
    m->Bmax = c.max_frames_per_call;
    // ---- vision tower
    add_f32(m, "vit.embeddings.class_embedding", D);
    add_linear(m, "vit.patch_embed", D, m->Kpe, {"vit.embeddings.patch_embedding.weight"}, D);   // K zero-padded to x64
    m->slots["vit.patch_embed"].Klogical = 3 * c.vit_patch * c.vit_patch;
    add_f32(m, "vit.embeddings.position_embedding.weight", m->S, D);
    add_f32(m, "vit.pre_layrnorm.weight", D); add_f32(m, "vit.pre_layrnorm.bias", D);
    for (int l = 0; l < c.vit_layers_run; ++l) {
        std::string p = "vit.encoder.layers." + std::to_string(l) + ".";
        add_linear(m, p + "qkv", 3 * D, D, {p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight"}, D);
        add_f32_fused(m, p + "qkv.bias", 3 * D, {p + "self_attn.q_proj.bias", p + "self_attn.k_proj.bias", p + "self_attn.v_proj.bias"}, D);
        add_linear(m, p + "out", D, D, {p + "self_attn.out_proj.weight"}, D);
        add_f32(m, p + "self_attn.out_proj.bias", D);
        add_linear(m, p + "fc1", c.vit_mlp, D, {p + "mlp.fc1.weight"}, c.vit_mlp);
        add_f32(m, p + "mlp.fc1.bias", c.vit_mlp);
        add_linear(m, p + "fc2", D, c.vit_mlp, {p + "mlp.fc2.weight"}, D);
        add_f32(m, p + "mlp.fc2.bias", D);
        add_f32(m, p + "layer_norm1.weight", D); add_f32(m, p + "layer_norm1.bias", D);
        add_f32(m, p + "layer_norm2.weight", D); add_f32(m, p + "layer_norm2.bias", D);
    }
    // ---- connector
    const int d = c.conn_d_model, di = c.conn_expand * d, R = c.conn_dt_rank, ds = c.conn_d_state;
    if (has_conn) {
    add_linear(m, "proj.pre", d, c.conn_mm_hidden, {"proj.pre_net.fc3.weight"}, d);
    add_f32(m, "proj.pre_net.fc3.bias", d);
    const std::string sp = "proj.mamba_model.ssms.0.";
    add_f32(m, sp + "norm.weight", d); add_f32(m, sp + "norm.bias", d);
    add_linear(m, "proj.in_proj", 2 * di, d, {sp + "mixer.in_proj.weight"}, 2 * di);
    add_f32(m, sp + "mixer.conv1d.weight", di, c.conn_d_conv); add_f32(m, sp + "mixer.conv1d.bias", di);
    add_linear(m, "proj.x_proj", R + 2 * ds, di, {sp + "mixer.x_proj.weight"}, R + 2 * ds);
    add_linear(m, "proj.dt_proj", di, R, {sp + "mixer.dt_proj.weight"}, di);
    add_f32(m, sp + "mixer.dt_proj.bias", di);
    add_f32(m, sp + "mixer.A_log", di, ds); add_f32(m, sp + "mixer.D", di);
    add_linear(m, "proj.out_proj", d, di, {sp + "mixer.out_proj.weight"}, d);
    add_f32(m, "proj.mamba_model.norm_fn.weight", d); add_f32(m, "proj.mamba_model.norm_fn.bias", d);
    add_linear(m, "proj.post", d, d, {"proj.post_net.fc3.weight"}, d);
    add_f32(m, "proj.post_net.fc3.bias", d);
    // ---- gate (V/O-only: q_proj / k_proj / embed_tokens are dead at seq-len 1 and ignored)
    const int gdh = c.gate_hidden / c.gate_heads;
    for (int l = 0; l < c.gate_layers; ++l) {
        std::string p = "proj.cls_net.cls_model.model.layers." + std::to_string(l) + ".";
        add_linear(m, p + "v", c.gate_kv_heads * gdh, d, {p + "self_attn.v_proj.weight"}, c.gate_kv_heads * gdh);
        add_linear(m, p + "o", d, c.gate_heads * gdh, {p + "self_attn.o_proj.weight"}, d);
        add_linear(m, p + "gu", 2 * c.gate_mlp, d, {p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"}, c.gate_mlp);
        add_linear(m, p + "down", d, c.gate_mlp, {p + "mlp.down_proj.weight"}, d);
        add_f32(m, p + "input_layernorm.weight", d); add_f32(m, p + "post_attention_layernorm.weight", d);
        m->ignored_prefixes.push_back(p + "self_attn.q_proj."); m->ignored_prefixes.push_back(p + "self_attn.k_proj.");
    }
    add_f32(m, "proj.cls_net.cls_model.model.norm.weight", d);
    add_linear(m, "proj.gate_head", 2, d, {"proj.cls_net.cls_model.lm_head.weight"}, 2);
    m->ignored_prefixes.push_back("proj.cls_net.cls_model.model.embed_tokens.");
    } else {
        m->ignored_prefixes.push_back("proj.");         // the projector's tensors belong to the host-side connector class
    }
    m->ignored_prefixes.push_back("vit.post_layernorm.");
    m->ignored_prefixes.push_back("vit.embeddings.position_ids");
    m->ignored_prefixes.push_back("vit.visual_projection.");        // CLIPVisionModelWithProjection / full CLIP directories
    // ---- LLM
    if (c.llm_layers > 0) {
        const int ld = c.llm_hidden, dh = ld / c.llm_heads, qn = c.llm_heads * dh, kn = c.llm_kv_heads * dh;
        SM_REQUIRE(qn == kn * (c.llm_heads / c.llm_kv_heads), "llm heads");
        Slot& e = m->slots["llm.embed"]; e.kind = 2; e.N = c.llm_vocab; e.K = ld;
        m->routes["llm.model.embed_tokens.weight"] = {"llm.embed", 0, c.llm_vocab, 0};
        for (int l = 0; l < c.llm_layers; ++l) {
            std::string p = "llm.model.layers." + std::to_string(l) + ".";
            // q, k, v fused; all three blocks must start on a 16-row boundary of the packed image
            SM_REQUIRE(qn % 16 == 0 && kn % 16 == 0, "llm q/k widths must be multiples of 16");
            Slot& s = m->slots[p + "qkv"]; s.kind = 1; s.N = qn + 2 * kn; s.K = ld; s.parts = 3;
            m->routes[p + "self_attn.q_proj.weight"] = {p + "qkv", 0, qn, 0};
            m->routes[p + "self_attn.k_proj.weight"] = {p + "qkv", qn, kn, 1};
            m->routes[p + "self_attn.v_proj.weight"] = {p + "qkv", qn + kn, kn, 2};
            add_linear(m, p + "o", ld, qn, {p + "self_attn.o_proj.weight"}, ld);
            add_linear(m, p + "gu", 2 * c.llm_mlp, ld, {p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"}, c.llm_mlp);
            add_linear(m, p + "down", ld, c.llm_mlp, {p + "mlp.down_proj.weight"}, ld);
            add_f32(m, p + "input_layernorm.weight", ld); add_f32(m, p + "post_attention_layernorm.weight", ld);
        }
        add_f32(m, "llm.model.norm.weight", ld);
        add_linear(m, "llm.lm_head", c.llm_vocab, ld, {"llm.lm_head.weight"}, c.llm_vocab);
    }
    if (c.vit_fp16)
        for (auto& kv : m->slots)
            if (kv.second.kind == 1 && kv.first.rfind("vit.", 0) == 0) kv.second.f16 = true;
    SM_REQUIRE(!(c.llm_fp16 && c.weights_fp8), "llm_fp16 and weights_fp8 are exclusive (fp16 operands exclude the fp8 weight-streaming kernels)");
    if (c.llm_fp16)      // the LLM's linear weights AND its embedding table are kept as IEEE fp16 (the reference loads its checkpoints that way, model/builder.py:54)
        for (auto& kv : m->slots)
            if ((kv.second.kind == 1 || kv.second.kind == 2) && kv.first.rfind("llm.", 0) == 0) kv.second.f16 = true;
    SM_REQUIRE(!(c.proj_fp16 && c.weights_fp8), "proj_fp16 and weights_fp8 are exclusive");
    if (c.proj_fp16)     // connector + gate linears in fp16 (their activations stay fp32-class: fp16 hi/lo pairs in gate_precise mode)
        for (auto& kv : m->slots)
            if (kv.second.kind == 1 && kv.first.rfind("proj.", 0) == 0) kv.second.f16 = true;
    if (c.weights_fp8)
        for (auto& kv : m->slots)
            if (kv.second.kind == 1 && (kv.first.rfind("proj.cls_net.", 0) == 0 || kv.first.rfind("llm.", 0) == 0 || kv.first == "proj.gate_head")) {
                kv.second.fp8 = true;
                kv.second.wo = kv.first.rfind("llm.", 0) != 0;
            }
    *out = m;
    return SM_OK;
}

// checkpoint name -> canonical "vit." / "proj." / "llm." name
static std::string canon(const std::string& name, const sm_config_t& c) {
    size_t p;
    if ((p = name.rfind("vision_tower.")) != std::string::npos) {
        std::string r = name.substr(p + 13);
        if (r.rfind("vision_model.", 0) == 0) r = r.substr(13);
        return "vit." + r;
    }
    if ((p = name.find("mm_projector.")) != std::string::npos) return "proj." + name.substr(p + 13);
    return "llm." + name;
}

extern "C" int sm_model_load_tensor(sm_model* m, const char* name_c, const void* data, int dtype, int ndim,
                                    const int64_t* shape, void* stream) {
    SM_REQUIRE(m && name_c && data && shape && ndim >= 1 && ndim <= 4, "sm_model_load_tensor: bad args");
    SM_REQUIRE(dtype == SM_DT_BF16 || dtype == SM_DT_F32 || dtype == SM_DT_F16, "sm_model_load_tensor: dtype must be bf16, f16 or f32");
    hipStream_t st = (hipStream_t)stream;
    std::string name = canon(name_c, m->c);
    auto it = m->routes.find(name);
    if (it == m->routes.end()) {
        for (auto& pre : m->ignored_prefixes)
            if (name.rfind(pre, 0) == 0) return 1;
        // vision layers beyond the selected hidden state are computed by HF but never read
        if (name.rfind("vit.encoder.layers.", 0) == 0) {
            int l = atoi(name.c_str() + 19);
            if (l >= m->c.vit_layers_run) return 1;
        }
        SM_FAIL(SM_EINVAL, "sm_model_load_tensor: unknown tensor '%s' (canonical '%s')", name_c, name.c_str());
    }
    Slot& s = m->slots[it->second.slot];
    const int row0 = it->second.row0;
    int64_t rows = shape[0], cols = 1;
    for (int i = 1; i < ndim; ++i) cols *= shape[i];
    const size_t n = (size_t)rows * cols;
    if (s.kind == 1 || s.kind == 2) {
        SM_REQUIRE(cols == (s.Klogical ? s.Klogical : s.K) && rows == it->second.rows,
                   "tensor '%s': shape [%lld x %lld] does not fit the expected [%d x %d]",
                   name_c, (long long)rows, (long long)cols, it->second.rows, s.Klogical ? s.Klogical : s.K);
        DevBuf tmp;
        const bf16_t* src = (const bf16_t*)data;
        const unsigned nb = (unsigned)((n + 255) / 256);
        if (s.f16) {     // fp16 image (vit_fp16 / llm_fp16): an fp16 checkpoint tensor goes in AS IS (builder.py:54 loads fp16)
            if (dtype != SM_DT_F16) {
                int rc = tmp.alloc(n * 2);
                if (rc) return rc;
                if (dtype == SM_DT_F32) f32_to_f16_kernel<<<nb, 256, 0, st>>>((const float*)data, tmp.as<bf16_t>(), n);
                else bf16_to_f16_kernel<<<nb, 256, 0, st>>>((const bf16_t*)data, tmp.as<bf16_t>(), n);
                src = tmp.as<bf16_t>();
            }
        } else if (dtype != SM_DT_BF16) {      // fp32 / fp16 (the reference loads fp16, model/builder.py:54) -> bf16, round to nearest even
            int rc = tmp.alloc(n * 2);
            if (rc) return rc;
            if (dtype == SM_DT_F32) f32_to_bf16_kernel<<<nb, 256, 0, st>>>((const float*)data, tmp.as<bf16_t>(), n);
            else f16_to_bf16_kernel<<<nb, 256, 0, st>>>((const _Float16*)data, tmp.as<bf16_t>(), n);
            src = tmp.as<bf16_t>();
        }
        if (s.kind == 1 && s.fp8) {
            if (!s.buf.p) {
                int rc = s.buf.alloc(sm_packed_fp8_bytes(s.N, s.K), true); if (rc) return rc;
                rc = s.scale.alloc((size_t)s.N * 4, true); if (rc) return rc;
            }
            SM_REQUIRE(row0 % 16 == 0, "fused part must start on a 16-row boundary");
            const int KSP = ((s.K + 31) / 32 + 1) / 2;
            char* dst = (char*)s.buf.p + (size_t)(row0 / 16) * KSP * 1024;
            int rc = sm_quant_pack_weight_fp8(src, (int)rows, s.K, (int)cols, dst, s.scale.as<float>() + row0, stream);
            if (rc) return rc;
        } else if (s.kind == 1) {
            if (!s.buf.p) { int rc = s.buf.alloc(sm_packed_elems(s.N, s.K) * 2, true); if (rc) return rc; }
            SM_REQUIRE(row0 % 16 == 0, "fused part must start on a 16-row boundary");
            const int KS = (s.K + 31) / 32;
            bf16_t* dst = s.buf.as<bf16_t>() + (size_t)(row0 / 16) * KS * 512;
            int rc = sm_pack_weight_ks(src, (int)rows, (int)cols, (int)cols, KS, dst, stream);
            if (rc) return rc;
        } else {
            if (!s.buf.p) { int rc = s.buf.alloc((size_t)s.N * s.K * 2); if (rc) return rc; }
            SM_HIP(hipMemcpyAsync(s.buf.as<bf16_t>() + (size_t)row0 * s.K, src, n * 2, hipMemcpyDeviceToDevice, st));
        }
        if (tmp.p) SM_HIP(hipStreamSynchronize(st));      // tmp is freed on return
    } else {
        SM_REQUIRE(n == (size_t)it->second.rows * s.K, "tensor '%s': %zu elements do not fit the expected %zu", name_c, n, (size_t)it->second.rows * s.K);
        if (!s.buf.p) { int rc = s.buf.alloc((size_t)s.N * s.K * 4); if (rc) return rc; }
        float* dst = s.buf.as<float>() + (size_t)row0 * s.K;
        if (dtype == SM_DT_F32) SM_HIP(hipMemcpyAsync(dst, data, n * 4, hipMemcpyDeviceToDevice, st));
        else if (dtype == SM_DT_F16) f16_to_f32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const _Float16*)data, dst, n);
        else bf16_to_f32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const bf16_t*)data, dst, n);
    }
    SM_LAUNCH_CHECK();
    s.loaded |= 1u << it->second.part;
    return SM_OK;
}

extern "C" int sm_model_missing(sm_model* m, char* buf, size_t buflen) {
    SM_REQUIRE(m && buf && buflen > 0, "sm_model_missing: bad args");
    std::string out;
    int cnt = 0;
    for (auto& kv : m->slots)
        if (!kv.second.complete()) { out += kv.first; out += "\n"; ++cnt; }
    snprintf(buf, buflen, "%s", out.c_str());
    return cnt;
}

static sm_linear_t lin(const sm_model* m, const Slot& w, const void* x, int x_dtype, int M, int ldx);
extern "C" int sm_model_finalize(sm_model* m, void* stream) {
    SM_REQUIRE(m, "sm_model_finalize: null model");
    for (auto& kv : m->slots)
        SM_REQUIRE(kv.second.complete(), "sm_model_finalize: tensor group '%s' incomplete (%d of %d parts)",
                   kv.first.c_str(), __builtin_popcount(kv.second.loaded), kv.second.parts);
    const sm_config_t& c = m->c;
    const int D = c.vit_hidden, B = m->Bmax, S = m->S;
    const size_t rows = (size_t)B * S;
    int rc;
    {   sm_model::VitWs* w0; if ((rc = m->vit_workspace(stream, &w0))) return rc; }       // the caller's stream: ready before the first hot call
    {   // resolve the names once
        auto S = [&](const std::string& n) { return (const Slot*)&m->slots.at(n); };
        auto F = [&](const std::string& n) { return (const float*)m->slots.at(n).buf.p; };
        sm_model::Resolved& R = m->R;
        R.patch = S("vit.patch_embed"); R.cls = F("vit.embeddings.class_embedding"); R.pos = F("vit.embeddings.position_embedding.weight");
        R.pre_ln_w = F("vit.pre_layrnorm.weight"); R.pre_ln_b = F("vit.pre_layrnorm.bias");
        R.vit.resize(c.vit_layers_run);
        for (int l = 0; l < c.vit_layers_run; ++l) {
            const std::string p = "vit.encoder.layers." + std::to_string(l) + ".";
            sm_model::LayerW& w = R.vit[l];
            w.qkv = S(p + "qkv"); w.out = S(p + "out"); w.fc1 = S(p + "fc1"); w.fc2 = S(p + "fc2");
            w.qkv_b = F(p + "qkv.bias"); w.out_b = F(p + "self_attn.out_proj.bias"); w.fc1_b = F(p + "mlp.fc1.bias"); w.fc2_b = F(p + "mlp.fc2.bias");
            w.ln1_w = F(p + "layer_norm1.weight"); w.ln1_b = F(p + "layer_norm1.bias"); w.ln2_w = F(p + "layer_norm2.weight"); w.ln2_b = F(p + "layer_norm2.bias");
        }
        const std::string sp = "proj.mamba_model.ssms.0.";
        if (c.conn_d_state > 0) {
        R.pre = S("proj.pre"); R.pre_b = F("proj.pre_net.fc3.bias"); R.cn_w = F(sp + "norm.weight"); R.cn_b = F(sp + "norm.bias");
        R.in_proj = S("proj.in_proj"); R.conv_w = F(sp + "mixer.conv1d.weight"); R.conv_b = F(sp + "mixer.conv1d.bias");
        R.x_proj = S("proj.x_proj"); R.dt_proj = S("proj.dt_proj"); R.dt_b = F(sp + "mixer.dt_proj.bias");
        R.A_log = F(sp + "mixer.A_log"); R.Dp = F(sp + "mixer.D"); R.out_proj = S("proj.out_proj");
        R.nf_w = F("proj.mamba_model.norm_fn.weight"); R.nf_b = F("proj.mamba_model.norm_fn.bias");
        R.post = S("proj.post"); R.post_b = F("proj.post_net.fc3.bias");
        R.gate.resize(c.gate_layers);
        for (int l = 0; l < c.gate_layers; ++l) {
            const std::string p = "proj.cls_net.cls_model.model.layers." + std::to_string(l) + ".";
            sm_model::LayerW& w = R.gate[l];
            w.v = S(p + "v"); w.o = S(p + "o"); w.gu = S(p + "gu"); w.down = S(p + "down");
            w.ln1_w = F(p + "input_layernorm.weight"); w.ln2_w = F(p + "post_attention_layernorm.weight");
        }
        R.gate_norm = F("proj.cls_net.cls_model.model.norm.weight"); R.gate_head = S("proj.gate_head");
        if (!R.gate_head->fp8 && R.gate_head->N == 2 && c.conn_d_model <= 8192 && (c.conn_d_model & 3) == 0) {
            if ((rc = m->gate_head_f32.alloc((size_t)2 * c.conn_d_model * sizeof(float)))) return rc;
            if ((rc = sm_unpack_rows_f32(R.gate_head->buf.p, 2, c.conn_d_model, R.gate_head->f16 ? 1 : 0, m->gate_head_f32.as<float>(), stream))) return rc;
        }
        }
        R.llm.resize(c.llm_layers);
        for (int l = 0; l < c.llm_layers; ++l) {
            const std::string p = "llm.model.layers." + std::to_string(l) + ".";
            sm_model::LayerW& w = R.llm[l];
            w.qkv = S(p + "qkv"); w.o = S(p + "o"); w.gu = S(p + "gu"); w.down = S(p + "down");
            w.ln1_w = F(p + "input_layernorm.weight"); w.ln2_w = F(p + "post_attention_layernorm.weight");
        }
        if (c.llm_layers > 0) { R.llm_norm = F("llm.model.norm.weight"); R.lm_head = S("llm.lm_head"); R.embed = S("llm.embed"); }
    }
    if (c.vit_hidden % 256 == 0 && c.vit_mlp % 256 == 0) {
        // LayerNorm folding (sm_linear_t.fold_*; vit_body_lanes uses it once a lane's products all run on the 256 x 256 tile kernels): g = W . gamma and
        // c = W . beta + b per folded LayerNorm, in fp32 -- the weight-streaming product with the fp32 vector entering as 16-bit hi + lo (2^-17 relative), once
        sm_model::Resolved& R = m->R;
        const int D = c.vit_hidden, F = c.vit_mlp;
        m->fold_vecs.resize((size_t)c.vit_layers_run * 4);
        for (int l = 0; l < c.vit_layers_run; ++l) {
            sm_model::LayerW& w = R.vit[l];
            const struct { const Slot* W; const float* gam; const float* bet; const float* bias; int N; const float** g; const float** cc; } jobs[2] = {
                {w.qkv, w.ln1_w, w.ln1_b, w.qkv_b, 3 * D, &w.qkv_g, &w.qkv_c}, {w.fc1, w.ln2_w, w.ln2_b, w.fc1_b, F, &w.fc1_g, &w.fc1_c}};
            for (int j = 0; j < 2; ++j) {
                DevBuf& gb = m->fold_vecs[(size_t)l * 4 + j * 2];
                DevBuf& cbuf = m->fold_vecs[(size_t)l * 4 + j * 2 + 1];
                if ((rc = gb.alloc((size_t)jobs[j].N * 4)) || (rc = cbuf.alloc((size_t)jobs[j].N * 4))) return rc;
                sm_linear_t a = lin(m, *jobs[j].W, jobs[j].gam, SM_X_F32, 1, D);
                a.precise = 1; a.out_f32 = gb.as<float>(); a.ldo = jobs[j].N;
                if ((rc = sm_linear(&a, stream))) return rc;
                sm_linear_t b = lin(m, *jobs[j].W, jobs[j].bet, SM_X_F32, 1, D);
                b.precise = 1; b.bias = jobs[j].bias; b.out_f32 = cbuf.as<float>(); b.ldo = jobs[j].N;
                if ((rc = sm_linear(&b, stream))) return rc;
                *jobs[j].g = gb.as<float>(); *jobs[j].cc = cbuf.as<float>();
            }
        }
    }
    m->finalized = true;
    return SM_OK;
}

extern "C" void sm_model_destroy(sm_model* m) { delete m; }
// fp8 models only: 1 = weight-only fp8 everywhere, 2 = fp8 x fp8 MFMA for calls with more than 16 rows (same weight images: the
// choice is made per call, so one model can be measured both ways)
extern "C" int sm_model_set_fp8_mode(sm_model* m, int mode) {
    SM_REQUIRE(m && m->c.weights_fp8 >= 1 && (mode == 1 || mode == 2), "sm_model_set_fp8_mode: needs a weights_fp8 model and mode 1 or 2");
    m->c.weights_fp8 = mode;
    return SM_OK;
}

// ------------------------------------------------------------------------------------------------ ViT
static sm_linear_t lin(const sm_model* m, const Slot& w, const void* x, int x_dtype, int M, int ldx) {
    sm_linear_t a;
    memset(&a, 0, sizeof(a));
    a.w = w.buf.p; a.N = w.N; a.K = w.K; a.x = x; a.x_dtype = x_dtype; a.M = M; a.ldx = ldx;
    if (w.fp8) { a.w_dtype = m->c.weights_fp8 == 2 && !w.wo ? SM_W_FP8_MFMA : SM_W_FP8; a.w_scale = w.scale.as<float>(); }
    if (w.f16) a.op_dtype = SM_OP_F16;
    return a;
}

extern "C" int sm_patchify_pixels(const void* pix, int dtype, int B, int H, int W, int patch, void* patches, int ldp, int op_dtype, void* stream);
extern "C" int sm_norm_ex(const float* x, int M, int D, int ldx, const float* gamma, const float* beta, float eps, int post_act,
                          float* out_f32, void* out_bf16, int ldo, int op_dtype, void* stream);
static int vit_body(sm_model* m, sm_model::VitWs* ws, int B, float* pooled, void* feats, void* stream);
// LayerNorm folding of the tower (vit_body_lanes): -2 = as SM_VIT_LN_FOLD / the default says, -1 = the default (fp16 operands fold, bf16 do not), 0 = off, 1 = on
static std::atomic<int> g_vit_ln_fold{-2};
extern "C" int sm_set_vit_ln_fold(int mode) {
    SM_REQUIRE(mode >= -2 && mode <= 1, "sm_set_vit_ln_fold: mode %d outside [-2, 1]", mode);
    g_vit_ln_fold.store(mode);
    return SM_OK;
}
struct VitLaneArgs { sm_model::VitWs* ws; int B; float* pooled; void* feats; void* stream; int f0 = 0; };   // f0: first frame slot of the lane inside `ws` (frame lanes of a small call share one workspace)
static int vit_body_lanes(sm_model* m, const VitLaneArgs* lanes, int nl);

// Two tower lanes: a call with >= SM_VIT_LANE_MIN frames (default 29: more than the 28 that fill the chip's 256 CUs with exactly one round of 256-row tiles; SM_VIT_LANES=1 disables) is cut in two halves that run the
// whole tower CONCURRENTLY, the first on the caller's stream, the second on a side stream of that caller stream (own
// workspaces; fork / join by events, nothing for the caller to do).  The tower alternates MFMA-bound GEMM main loops with
// HBM-bound phases (fp32-residual epilogues, LayerNorm, pooling) and a 256x256 GEMM block leaves no room for a second resident
// block, so a single batch runs them strictly one after the other; two independent half batches fill each other's epilogues,
// kernel tails and launch gaps (+8 % frames/s at 2 x 28 frames, tools/two_stream_bench.py).  Results are those of two calls.
static int vit_lane_count(int B) {
    static int lanes = -1, min_b = 29;      // measured: 29 frames 1842 -> 2220 frames/s, 32: 1936 -> 2113, 40: 2001 -> 2104, 44: 2069 -> 2253; 28 as 2 x 14: 2255 -> 2209
    if (lanes < 0) {
        const char* e = getenv("SM_VIT_LANES"); lanes = e ? atoi(e) : 2;
        const char* mb = getenv("SM_VIT_LANE_MIN"); if (mb) min_b = atoi(mb);
    }
    return lanes >= 2 && B >= min_b ? 2 : 1;
}
// Frame lanes of a SMALL call (2 .. 8 frames: the reference's own operating point is a tick of one or a few new frames).
// One frame through the tower is ~185 dependent launches of 5-15 us, most of it the fixed cost of a dependent launch (a kernel that
// does nothing takes 4.7 us between two others; profiles/r04_tick_b1_timeline.txt) and none of them fills the chip (577 rows = 40-160
// tiles of 128 x 128): the frames of a small call are independent, so each gets its own HIP stream and the launch chains run side by
// side (same workspace, disjoint frame slots; fork / join by events like the two big lanes).  Results are those of one-frame calls.
// MEASURED AND LEFT OFF (SM_VIT_SMALL_LANES=n enables n lanes): 4 frames as 4 lanes 5.72 ms against 3.74 ms as one batch, 8 frames as 8
// lanes 9.9 vs 5.56 -- and with one HOST THREAD per lane (tools/lanes_threads_probe.py: the issue cost taken out) 4 frames still take
// 3.52 ms: the chip retires ~one dependent launch per 4.7 us ACROSS all queues (740 launches of four concurrent one-frame chains in
// 3.5 ms), so side-by-side launch chains do not buy latency here; only fewer launches do.
// ROUND 6 -- TWO frame lanes where one lane sits between whole rounds of the 128 x 128 tiles (the 7 -> 8 and 14 -> 16 frame steps of the per-call latency; kernel traces in
// profiles/r06_tick_cliff_kernels.txt).  Out-proj and fc2 are rows / 128 x 8 tiles: up to 256 of them run one per CU on the ring kernel (7 frames, 30 us), 296 (8 frames) go
// to the two-stage kernel with two blocks on 40 of the CUs and take as long as those 40 do (46 us); 512 is two blocks everywhere, 584 (16 frames) a second round of 72.
// Two half batches on two HIP streams put the same tiles on the chip as two launches of half the size whose tails fill each other: measured per frame count, same box, twice
// (profiles/r06_small_lanes_scan.txt), 8 / 9 / 10 frames 5.47 / 5.78 / 6.22 -> 5.05 / 5.43 / 5.77 ms, 15 .. 20 frames -4 / -8 / -9 / -5 / -5 / -2 %, and a loss of 1-8 % everywhere
// else (4 .. 7, 11, 12, 14, 21 .. 28 frames) -- hence a rule in tiles, not a switch.  Results are those of the two half calls (tested).
static std::atomic<int> g_vit_frame_lanes{-2};          // sm_set_vit_frame_lanes: -2 = read SM_VIT_SMALL_LANES (default -1)
extern "C" int sm_set_vit_frame_lanes(int mode) {
    SM_REQUIRE(mode >= -2 && mode <= 8 && mode != 0, "sm_set_vit_frame_lanes: -2 (environment / default), -1 (the tile rule), 1 (never), 2..8 (that many lanes up to SM_VIT_SMALL_MAX frames)");
    g_vit_frame_lanes.store(mode, std::memory_order_relaxed);
    return SM_OK;
}
static int vit_small_lanes(int B, int S, int D) {
    static int max_b = -1;
    int max_l = g_vit_frame_lanes.load(std::memory_order_relaxed);
    if (max_l == -2) {
        const char* e = getenv("SM_VIT_SMALL_LANES"); max_l = e ? atoi(e) : -1;          // -1: the tile rule below; 1: never; n >= 2: n lanes up to SM_VIT_SMALL_MAX frames (the round-4 experiment)
        if (max_l > 8) max_l = 8;
        if (max_l == 0 || max_l < -1) max_l = 1;
        g_vit_frame_lanes.store(max_l, std::memory_order_relaxed);
    }
    if (max_b < 0) { const char* mb = getenv("SM_VIT_SMALL_MAX"); max_b = mb ? atoi(mb) : 8; }
    if (max_l < 0) {
        static int n_cu = 0;
        if (!n_cu) { int dev = 0; hipDeviceProp_t prop; n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256; }
        const long t = (long)cdiv(B * S, 128) * cdiv(D, 128);            // 128 x 128 tiles of the N = vit_hidden products (out-proj, fc2) of ONE lane: rows / 128 x 8 for ViT-L
        const bool between = (t > n_cu && 2 * t <= 3 * n_cu) || (t > 2 * n_cu && 8 * t <= 23 * n_cu);       // (256, 384] and (512, 736] tiles on 256 CUs
        return B >= 2 && between ? 2 : 1;
    }
    if (B < 2 || B > max_b || max_l < 2) return 1;
    return B < max_l ? B : max_l;
}
template <class Front>      // front(patches, first frame, frames, stream): fills the patch matrix `patches` for `frames` frames starting at `first frame`
static int vit_encode_lanes(sm_model* m, int B, float* pooled, void* feats, void* stream, Front front) {
    const sm_config_t& c = m->c;
    sm_model::VitWs* ws;
    int rc = m->vit_workspace(stream, &ws);
    if (rc) return rc;
    if (const int sl = vit_small_lanes(B, m->S, c.vit_hidden); sl > 1) {
        sm_model::Lane* L;
        if ((rc = m->lane_of(stream, &L, sl - 1))) return rc;
        VitLaneArgs args[8];
        SM_HIP(hipEventRecord(L->fork, (hipStream_t)stream));
        for (int i = 0; i < sl; ++i) {
            const int f0 = (int)((long)B * i / sl), f1 = (int)((long)B * (i + 1) / sl);
            void* st = i == 0 ? stream : (void*)L->more[i - 1];
            if (i) SM_HIP(hipStreamWaitEvent((hipStream_t)st, L->fork, 0));
            args[i] = VitLaneArgs{ws, f1 - f0, pooled + (size_t)f0 * c.vit_hidden, feats ? (char*)feats + (size_t)f0 * m->P * c.vit_hidden * 2 : nullptr, st, f0};
            if ((rc = front((char*)ws->patches.p + (size_t)f0 * m->P * m->Kpe * 2, f0, f1 - f0, st))) return rc;
        }
        if ((rc = vit_body_lanes(m, args, sl))) return rc;
        for (int i = 1; i < sl; ++i) {
            SM_HIP(hipEventRecord(L->more_join[i - 1], L->more[i - 1]));
            SM_HIP(hipStreamWaitEvent((hipStream_t)stream, L->more_join[i - 1], 0));
        }
        return SM_OK;
    }
    if (vit_lane_count(B) == 1) {
        if ((rc = front(ws->patches.p, 0, B, stream))) return rc;
        return vit_body(m, ws, B, pooled, feats, stream);
    }
    sm_model::Lane* L;
    sm_model::VitWs* ws1;
    if ((rc = m->lane_of(stream, &L))) return rc;
    if ((rc = m->vit_workspace((void*)L->side, &ws1))) return rc;
    // lanes of at most `cap` frames = one round of 256-row tiles on this chip for the N = 1024 GEMMs (28 frames of 577 tokens on 256
    // CUs), balanced, two at a time: 56 frames -> 2 x 28, 84 -> 3 x 28 (the third alone), 100 -> 4 x 25.  Measured 2 x 56 against
    // 2 x (2 x 28): 2379 vs 2418 frames/s.
    int n_cu = 256;
    { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount; }
    int cap = (n_cu / 4) * 256 / m->S;
    if (cap < 1) cap = 1;
    const int nl = B <= 2 * cap ? 2 : cdiv(B, cap);
    auto first = [&](int i) { return (int)((long)B * i / nl); };          // lane i covers frames [first(i), first(i + 1))
    SM_HIP(hipEventRecord(L->fork, (hipStream_t)stream));            // the frames (and the output buffers' previous readers) are in stream order
    SM_HIP(hipStreamWaitEvent(L->side, L->fork, 0));
    for (int i = 0; i < nl; i += 2) {
        const int f0 = first(i), f1 = first(i + 1), f2 = i + 1 < nl ? first(i + 2) : f1;
        VitLaneArgs two[2] = {{ws, f1 - f0, pooled + (size_t)f0 * c.vit_hidden, feats ? (char*)feats + (size_t)f0 * m->P * c.vit_hidden * 2 : nullptr, stream},
                              {ws1, f2 - f1, pooled + (size_t)f1 * c.vit_hidden, feats ? (char*)feats + (size_t)f1 * m->P * c.vit_hidden * 2 : nullptr, (void*)L->side}};
        const int n2 = f2 > f1 ? 2 : 1;
        if ((rc = front(ws->patches.p, f0, f1 - f0, stream))) return rc;
        if (n2 == 2 && (rc = front(ws1->patches.p, f1, f2 - f1, (void*)L->side))) return rc;
        if ((rc = vit_body_lanes(m, two, n2))) return rc;
    }
    SM_HIP(hipEventRecord(L->join, L->side));
    SM_HIP(hipStreamWaitEvent((hipStream_t)stream, L->join, 0));
    return SM_OK;
}

extern "C" int sm_vit_encode(sm_model* m, const uint8_t* frames, int B, float* pooled, void* feats, float* pix, void* stream) {
    SM_REQUIRE(m && m->finalized, "sm_vit_encode: model not finalized");
    SM_REQUIRE(frames && pooled && B >= 1 && B <= m->Bmax, "sm_vit_encode: B=%d outside [1, %d]", B, m->Bmax);
    const sm_config_t& c = m->c;
    const size_t fpx = (size_t)c.vit_image * c.vit_image * 3;
    // a1: u8 ring buffer -> normalised bf16 patch matrix
    return vit_encode_lanes(m, B, pooled, feats, stream, [&](void* patches, int f0, int nf, void* st) {
        return sm_preprocess_patches(frames + (size_t)f0 * fpx, nf, c.vit_image, c.vit_image, c.vit_patch, c.img_mean, c.img_std, patches, m->Kpe,
                                     pix ? pix + (size_t)f0 * fpx : nullptr, c.vit_fp16 ? SM_OP_F16 : SM_OP_BF16, st);
    });
}

extern "C" int sm_vit_encode_pixels(sm_model* m, const void* pixel_values, int dtype, int B, float* pooled, void* feats, void* stream) {
    SM_REQUIRE(m && m->finalized, "sm_vit_encode_pixels: model not finalized");
    SM_REQUIRE(pixel_values && pooled && B >= 1 && B <= m->Bmax, "sm_vit_encode_pixels: B=%d outside [1, %d]", B, m->Bmax);
    const sm_config_t& c = m->c;
    const size_t fpx = (size_t)c.vit_image * c.vit_image * 3 * (dtype == SM_DT_F32 ? 4 : 2);
    return vit_encode_lanes(m, B, pooled, feats, stream, [&](void* patches, int f0, int nf, void* st) {
        return sm_patchify_pixels((const char*)pixel_values + (size_t)f0 * fpx, dtype, nf, c.vit_image, c.vit_image, c.vit_patch, patches, m->Kpe,
                                  c.vit_fp16 ? SM_OP_F16 : SM_OP_BF16, st);
    });
}

// the tower over `nl` independent half batches, issued INTERLEAVED (every step for lane 0, then the same step for lane 1): both
// lanes start together and stay a step apart instead of one running ~2 ms (the host's issue time of a whole lane) behind the other
static int vit_body_lanes(sm_model* m, const VitLaneArgs* lanes, int nl) {
    const sm_config_t& c = m->c;
    const sm_model::Resolved& R = m->R;
    const int D = c.vit_hidden, H = c.vit_heads, dh = D / H, S = m->S, P = m->P;
    int rc;
    const int od = c.vit_fp16 ? SM_OP_F16 : SM_OP_BF16;      // 16-bit type of every ViT GEMM operand (weights are packed to match)
    static const bool fc2_mean = [] { const char* e = getenv("SM_VIT_FC2_MEAN"); return !e || atoi(e) != 0; }();
    // tile experiments for the two fp32 + residual shapes (SM_TILE_* values; 0 = sm_linear's own choice)
    static const int out_tile = [] { const char* e = getenv("SM_VIT_OUT_TILE"); return e ? atoi(e) : 0; }();
    static const int fc2_tile = [] { const char* e = getenv("SM_VIT_FC2_TILE"); return e ? atoi(e) : 0; }();
#define LANES for (int li = 0; li < nl; ++li)
#define LV const VitLaneArgs& L = lanes[li]; const size_t r0 = (size_t)L.f0 * S; float* x = L.ws->x.as<float>() + r0 * D; bf16_t* xn = L.ws->xn.as<bf16_t>() + r0 * D; \
    char* w_patches = (char*)L.ws->patches.p + (size_t)L.f0 * P * m->Kpe * 2; char* w_qkv = (char*)L.ws->qkv.p + r0 * 3 * D * 2; char* w_ctx = (char*)L.ws->ctx.p + r0 * D * 2; \
    char* w_hmid = (char*)L.ws->hmid.p + r0 * c.vit_mlp * 2; const int M = L.B * S; void* stream = L.stream; (void)x; (void)xn; (void)M; (void)w_patches; (void)w_qkv; (void)w_ctx; (void)w_hmid
    // patch-embed GEMM (+ position embedding) into token rows 1..P of every frame; CLS row; pre_layrnorm in place
    LANES { LV;
        sm_linear_t a = lin(m, *R.patch, w_patches, SM_X_BF16, L.B * P, m->Kpe);
        a.out_f32 = x; a.ldo = D;
        a.residual = R.pos; a.ldr = D;
        a.remap_in = P; a.remap_out = S; a.remap_off = 1;
        if ((rc = sm_linear(&a, stream))) return rc;
        if ((rc = sm_vit_cls_rows(x, L.B, S, D, R.cls, R.pos, stream))) return rc;
        if ((rc = sm_norm(x, M, D, D, R.pre_ln_w, R.pre_ln_b, c.vit_eps, 0, x, nullptr, D, stream))) return rc;
    }
    // LayerNorms ride behind the residual products (sm_linear_t.post_ln_*): out-proj leaves LN2(x), fc2 leaves the NEXT layer's LN1(x),
    // as the 16-bit operand `xn` of the following product -- fused into the slab pass for a single frame, a separate launch otherwise.
    // Only the first layer's LN1 is a call of its own.
    if (c.vit_layers_run > 0) LANES { LV;
        if ((rc = sm_norm_ex(x, M, D, D, R.vit[0].ln1_w, R.vit[0].ln1_b, c.vit_eps, 0, nullptr, xn, D, od, stream))) return rc;
    }
    // LayerNorm FOLDING (sm_linear_t.fold_*; SM_VIT_LN_FOLD=0 switches it off): once a lane's products all run on the 256 x 256 tile kernels (>= 192
    // tiles for the narrowest one, out-proj: >= 21 frames of 577 tokens), the LayerNorm in front of fc1 (every layer) and of q|k|v (layers >= 1) is no
    // launch of its own -- out-proj / fc2 leave 16-bit(x * gamma) + per-tile row sums beside the fp32 stream, q|k|v / fc1 apply mean and 1/std on their
    // accumulators: 45 of the 47 LayerNorm launches of a tower pass (100 MB each at 28 frames) go; layer 0's follows the pre-LayerNorm kernel.
    // DEFAULT: the fp16 tower folds, the bf16 tower does not.  Measured at full size on 28 frames (tools/fullsize_parity_probe.py, profiles/r06_fold_parity_probe.txt):
    // fp16 operands -- gate logits 1.7e-4 from the fp32 oracle (bound 1e-3), 2.4e-4 from the oracle mode that restates the fold; bf16 operands -- 2.5e-3
    // from fp32 (the dtype's floor, as without the fold: 2.3e-3) but 1.20e-3 from ITS restatement where the unfolded path sits at 8.8e-4: beyond the 1e-3
    // this build asserts for the benchmarked dtype against the matching-precision oracle (two equivalent bf16 towers differ by ~1e-3 on these logits, so
    // which side a variant lands on is luck -- round 3's form of the fold landed at 1.21e-3 too).  SM_VIT_LN_FOLD=1 folds both, =0 neither (A/B).
    // sm_set_vit_ln_fold(mode) overrides the environment at run time (tests, the bench's A/B leg).
    static const int fold_env = [] { const char* e = getenv("SM_VIT_LN_FOLD"); return e ? atoi(e) : -1; }();
    const int fold_mode = g_vit_ln_fold.load() >= -1 ? g_vit_ln_fold.load() : fold_env;
    const bool fold_on = fold_mode < 0 ? c.vit_fp16 != 0 : fold_mode != 0;
    auto folds = [&](int M) { return fold_on && D % 256 == 0 && c.vit_mlp % 256 == 0 && R.vit[0].qkv_g && out_tile == 0 && fc2_tile == 0 && cdiv(M, 256) * (D / 256) >= 192; };
    for (int l = 0; l < c.vit_layers_run; ++l) {
        const sm_model::LayerW& w = R.vit[l];
        const sm_model::LayerW* wnext = l + 1 < c.vit_layers_run ? &R.vit[l + 1] : nullptr;
        LANES { LV;
            sm_linear_t a = lin(m, *w.qkv, xn, SM_X_BF16, M, D);
            a.bias = w.qkv_b;
            a.out_bf16 = w_qkv; a.ldo_bf16 = 3 * D;
            if (folds(M) && l >= 1) { a.fold_stats_in = L.ws->stats.as<float>() + r0 * (D / 256) * 2; a.fold_g = w.qkv_g; a.fold_c = w.qkv_c; a.fold_eps = c.vit_eps; }
            if ((rc = sm_linear(&a, stream))) return rc;
        }
        // V is transposed inside the attention kernel's LDS staging (a V^T side output of the QKV GEMM cost ~50 us of
        // scalar 2-byte stores per layer at 28 frames)
        LANES { LV;
            if ((rc = sm_vit_attention(w_qkv, nullptr, w_ctx, L.B, S, H, dh, 0, od, stream))) return rc;
            sm_linear_t a = lin(m, *w.out, w_ctx, SM_X_BF16, M, D);
            a.bias = w.out_b;
            a.residual = x; a.ldr = D; a.out_f32 = x; a.ldo = D;
            a.tile_hint = out_tile;
            a.post_ln_gamma = w.ln2_w; a.post_ln_beta = w.ln2_b; a.post_ln_eps = c.vit_eps; a.post_ln_out = xn; a.post_ln_ldo = D;
            if (folds(M)) a.fold_stats_out = L.ws->stats.as<float>() + r0 * (D / 256) * 2;       // producer of layer_norm2's operand: xn = 16-bit(x * gamma2), row sums
            if ((rc = sm_linear(&a, stream))) return rc;
        }
        LANES { LV;
            sm_linear_t a = lin(m, *w.fc1, xn, SM_X_BF16, M, D);
            a.bias = w.fc1_b; a.act = SM_ACT_QUICK_GELU;
            a.out_bf16 = w_hmid; a.ldo_bf16 = c.vit_mlp;
            if (folds(M)) { a.fold_stats_in = L.ws->stats.as<float>() + r0 * (D / 256) * 2; a.fold_g = w.fc1_g; a.fold_c = w.fc1_c; a.fold_eps = c.vit_eps; }
            if ((rc = sm_linear(&a, stream))) return rc;
        }
        LANES { LV;
            // LAST layer, pooled feature only (the streaming path): pooled = mean_p(x_p + fc2(h_p) + b) = mean_p x_p + fc2(mean_p h_p) + b -- the
            // 135 GFLOP GEMM of a 28-frame lane becomes a patch mean over h (one pass over 132 MB) and a 28-row weight-streaming product
            // (exact algebra, fp32 means).  Not when the per-patch features are asked for.  SM_VIT_FC2_MEAN=0: the GEMM (A/B).
            if (!wnext && fc2_mean && !L.feats && L.B <= 32) {
                float* hbar = L.ws->hbar.as<float>() + (size_t)L.f0 * c.vit_mlp;
                if ((rc = sm_pool_patches16(w_hmid, L.B, S, c.vit_mlp, hbar, c.vit_fp16 ? 1 : 0, stream))) return rc;
                if ((rc = sm_pool_patches(x, L.B, S, D, L.pooled, nullptr, stream))) return rc;
                sm_linear_t a = lin(m, *w.fc2, hbar, SM_X_F32, L.B, c.vit_mlp);
                a.precise = 1; a.bias = w.fc2_b;
                a.residual = L.pooled; a.ldr = D; a.out_f32 = L.pooled; a.ldo = D;
                if ((rc = sm_linear(&a, stream))) return rc;
                continue;
            }
            sm_linear_t a = lin(m, *w.fc2, w_hmid, SM_X_BF16, M, c.vit_mlp);
            a.bias = w.fc2_b;
            a.residual = x; a.ldr = D; a.out_f32 = x; a.ldo = D;
            a.tile_hint = fc2_tile;
            if (wnext) { a.post_ln_gamma = wnext->ln1_w; a.post_ln_beta = wnext->ln1_b; a.post_ln_eps = c.vit_eps; a.post_ln_out = xn; a.post_ln_ldo = D; }
            if (wnext && folds(M)) a.fold_stats_out = L.ws->stats.as<float>() + r0 * (D / 256) * 2;      // producer of the NEXT layer's layer_norm1 operand
            if ((rc = sm_linear(&a, stream))) return rc;
        }
    }
    LANES { LV;
        if (c.vit_layers_run > 0 && fc2_mean && !L.feats && L.B <= 32) continue;         // pooled already written by the last layer (above)
        if ((rc = sm_pool_patches(x, L.B, S, D, L.pooled, L.feats, stream))) return rc;
    }
#undef LANES
#undef LV
    return SM_OK;
}
static int vit_body(sm_model* m, sm_model::VitWs* ws, int B, float* pooled, void* feats, void* stream) {
    const VitLaneArgs one = {ws, B, pooled, feats, stream};
    return vit_body_lanes(m, &one, 1);
}

// ------------------------------------------------------------------------------------------------ stream
#define SM_DECODE_SPLITS 32
// connector + gate scratch of one weight pass (<= 32 rows); owned by a stream, or by a stream group
struct ConnScratch {
    DevBuf pooled, t0, u, xz, xc, xdbl, delta, y, r, lnf, tokrows, h, hn, v, vrep, act, hfin, logits2;
    int alloc(const sm_model* m) {
        const sm_config_t& c = m->c;
        const int d = c.conn_d_model, di = c.conn_expand * d, R = c.conn_dt_rank, ds = c.conn_d_state;
        const int gdh = c.gate_hidden / c.gate_heads;
        const int xd = cdiv(R + 2 * ds, 32) * 32 + 32;
        int rc = 0;
#define A(buf, bytes, z) if (!rc) rc = buf.alloc((bytes), z)
        A(pooled, (size_t)(m->Bmax > 32 ? m->Bmax : 32) * c.conn_mm_hidden * 4, false);
        A(t0, (size_t)32 * d * 4, false); A(u, (size_t)32 * d * 4, false);
        A(xz, (size_t)32 * 2 * di * 4, false); A(xc, (size_t)32 * di * 4, false);
        A(xdbl, (size_t)32 * xd * 4, true); A(delta, (size_t)32 * di * 4, false);
        A(y, (size_t)32 * di * 4, false); A(r, (size_t)32 * d * 4, false); A(lnf, (size_t)32 * d * 4, false);
        A(tokrows, (size_t)32 * d * 4, false);
        A(h, (size_t)32 * d * 4, false); A(hn, (size_t)32 * d * 4, false);
        A(v, (size_t)32 * c.gate_kv_heads * gdh * 4, false); A(vrep, (size_t)32 * c.gate_heads * gdh * 4, false);
        A(act, (size_t)32 * c.gate_mlp * 4, false); A(hfin, (size_t)32 * d * 4, false);
        A(logits2, (size_t)32 * 2 * 4, false);
#undef A
        return rc;
    }
};

struct sm_stream {
    sm_model* m;
    int max_frames, max_seq;
    int T = 0, kv_len = 0;
    DevBuf conv_state, ssm_state, tokens;
    ConnScratch w;
    // LLM
    // K [cap][KV * dh] and V^T [KV * dh][cap] per layer.  `cap` (tokens, a multiple of 256) GROWS with the context up to max_seq (kv_reserve below): a stream
    // opened for 4096 tokens holds 512 until it needs more, so 512 open streams of a few hundred tokens are 32 GB of cache, not 256 GB
    std::vector<DevBuf> kc, vtc;
    int cap = 0;
    DevBuf lmlog, next_tok;
    // pipelined perception (sm_stream_push_frames_pipelined): the connector + gate pass of call i runs on this stream's own
    // side HIP stream while the caller's stream already runs the tower of call i+1
    hipStream_t side = nullptr;
    hipEvent_t ev_vit = nullptr, ev_pass[2] = {nullptr, nullptr};
    DevBuf pooled_pp[2];
    int flip = 0, last = -1;           // last: index of the event of the newest pass (-1: none yet)
    long pass_seq = 0, joined_seq = -1;   // number of the newest pass / of the pass `joined_on` was last ordered behind
    void* joined_on = nullptr;
    ~sm_stream() {
        if (side) { (void)hipStreamSynchronize(side); release_stream_workspaces(side); (void)hipStreamDestroy(side); }
        if (ev_vit) (void)hipEventDestroy(ev_vit);
        for (auto e : ev_pass) if (e) (void)hipEventDestroy(e);
    }
};

// every entry point that reads or writes a stream's state on the caller's HIP stream first orders that stream behind the
// pipelined passes still running on the side stream (a no-op when none is pending)
// (the newest pass stays "pending" for every OTHER HIP stream that comes along -- e.g. an LLM lane next to the perception stream --,
// only a repeated join of the same pass from the same stream is skipped)
static int auto_join(sm_stream* s, void* stream) {
    if (s && s->last >= 0 && !(s->joined_seq == s->pass_seq && s->joined_on == stream)) {
        SM_HIP(hipStreamWaitEvent((hipStream_t)stream, s->ev_pass[s->last], 0));
        s->joined_seq = s->pass_seq; s->joined_on = stream;
    }
    return SM_OK;
}

// ---- growable K / V cache.  SM_KV_INITIAL_CAP (tokens, default 512) is what a stream holds when it is opened; kv_reserve grows it geometrically (x2,
// whole multiples of 256, never beyond max_seq) before a call that needs more: new buffers, the live part of the old ones copied on the caller's HIP stream
// (K rows as they are; V^T row by row into the wider pitch), the stream drained, the old buffers freed.  Three growths take a stream from 512 to 4096 tokens,
// so the copy is amortised to less than one extra pass over the cache.  Not capturable (allocation + synchronisation): a captured step must find its
// capacity in place -- it does whenever the call before the capture ran eagerly at the same context.
static int kv_initial_cap(int max_seq) {
    static int init = -1;
    if (init < 0) { const char* e = getenv("SM_KV_INITIAL_CAP"); init = e ? atoi(e) : 512; if (init < 256) init = 256; }
    int cap = (init + 255) / 256 * 256;
    return cap < max_seq ? cap : max_seq;
}
static int kv_set_cap(sm_stream* s, int new_cap, void* stream) {
    const sm_config_t& c = s->m->c;
    if (new_cap <= s->cap) return SM_OK;
    SM_REQUIRE(new_cap <= s->max_seq && (new_cap % 64) == 0, "kv_reserve: capacity %d outside (cap, max_seq = %d] or not a multiple of 64", new_cap, s->max_seq);
    hipStream_t st = (hipStream_t)stream;
    {   hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess) SM_REQUIRE(cs == hipStreamCaptureStatusNone, "kv_reserve: the K / V cache must grow (%d -> %d tokens) but the HIP stream is capturing: run the step eagerly once at this context first", s->cap, new_cap);
        else (void)hipGetLastError(); }
    const int dh = c.llm_hidden / c.llm_heads, kn = c.llm_kv_heads * dh;
    std::vector<DevBuf> nk(c.llm_layers), nv(c.llm_layers);
    int rc = 0;
    for (int l = 0; l < c.llm_layers && !rc; ++l) {
        rc = nk[l].alloc((size_t)new_cap * kn * 2, false);
        if (!rc) rc = nv[l].alloc((size_t)kn * new_cap * 2, false);
    }
    if (rc) return rc;
    const int live = s->kv_len;
    for (int l = 0; l < c.llm_layers; ++l) {
        SM_HIP(hipMemsetAsync(nk[l].p, 0, (size_t)new_cap * kn * 2, st));
        SM_HIP(hipMemsetAsync(nv[l].p, 0, (size_t)kn * new_cap * 2, st));
        if (live > 0) {
            SM_HIP(hipMemcpyAsync(nk[l].p, s->kc[l].p, (size_t)live * kn * 2, hipMemcpyDeviceToDevice, st));
            SM_HIP(hipMemcpy2DAsync(nv[l].p, (size_t)new_cap * 2, s->vtc[l].p, (size_t)s->cap * 2, (size_t)live * 2, kn, hipMemcpyDeviceToDevice, st));
        }
    }
    SM_HIP(hipStreamSynchronize(st));            // the old buffers may still be read by kernels enqueued before this call
    for (int l = 0; l < c.llm_layers; ++l) { s->kc[l] = std::move(nk[l]); s->vtc[l] = std::move(nv[l]); }
    s->cap = new_cap;
    return SM_OK;
}
// make room for `need` tokens (kv_len + the rows of the call about to be issued)
static int kv_reserve(sm_stream* s, int need, void* stream) {
    if (need <= s->cap) return SM_OK;
    int nc = s->cap;
    while (nc < need) nc *= 2;
    nc = (nc + 255) / 256 * 256;
    if (nc > s->max_seq) nc = s->max_seq;
    return kv_set_cap(s, nc, stream);
}
extern "C" int sm_stream_kv_capacity(sm_stream* s) { return s ? s->cap : -1; }

extern "C" int sm_stream_open(sm_model* m, int max_frames, int max_seq, sm_stream** out) {
    SM_REQUIRE(m && m->finalized && out && max_frames > 0, "sm_stream_open: bad args / model not finalized");
    const sm_config_t& c = m->c;
    sm_stream* s = new sm_stream();
    s->m = m; s->max_frames = max_frames;
    const int d = c.conn_d_model, di = c.conn_expand * d, ds = c.conn_d_state;
    int rc = 0;
#define A(buf, bytes, z) if (!rc) rc = s->buf.alloc((bytes), z)
    A(conv_state, (size_t)di * c.conn_d_conv * 4, true);
    A(ssm_state, (size_t)di * ds * 4, true);
    A(tokens, (size_t)max_frames * d * 4, false);
    if (!rc) rc = s->w.alloc(m);
    if (!rc && c.llm_layers > 0) {
        SM_REQUIRE(max_seq > 0 && max_seq % 64 == 0, "sm_stream_open: max_seq must be a positive multiple of 64");
        s->max_seq = max_seq;
        const int ld = c.llm_hidden, dh = ld / c.llm_heads, kn = c.llm_kv_heads * dh, qn = c.llm_heads * dh;
        s->kc.resize(c.llm_layers); s->vtc.resize(c.llm_layers);
        s->cap = kv_initial_cap(max_seq);
        for (int l = 0; l < c.llm_layers && !rc; ++l) {
            rc = s->kc[l].alloc((size_t)s->cap * kn * 2, true);
            if (!rc) rc = s->vtc[l].alloc((size_t)kn * s->cap * 2, true);
        }
        (void)qn;
        A(lmlog, (size_t)c.llm_vocab * 4, false); A(next_tok, 64, true);
        if (!rc && m->rope_len < max_seq) {
            // cos/sin(pos * theta^(-2j/dh)) exactly as HF MistralRotaryEmbedding: fp32 inv_freq, fp32 product
            const int half = dh / 2;
            std::vector<float> hc((size_t)max_seq * half), hs((size_t)max_seq * half);
            for (int j = 0; j < half; ++j) {
                float inv = 1.0f / powf(c.llm_rope_theta, (float)(2 * j) / (float)dh);
                for (int p = 0; p < max_seq; ++p) {
                    float ang = (float)p * inv;
                    hc[(size_t)p * half + j] = (float)cos((double)ang);
                    hs[(size_t)p * half + j] = (float)sin((double)ang);
                }
            }
            rc = m->rope_cos.alloc(hc.size() * 4);
            if (!rc) rc = m->rope_sin.alloc(hs.size() * 4);
            if (!rc) {
                SM_HIP(hipMemcpy(m->rope_cos.p, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
                SM_HIP(hipMemcpy(m->rope_sin.p, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
                m->rope_len = max_seq;
            }
        }
    }
#undef A
    if (rc) { delete s; return rc; }
    *out = s;
    return SM_OK;
}

// the recurrent state is zeroed by a KERNEL, not hipMemsetAsync: captured into a hipGraph (BASELINE configs[4]'s captured per-frame step), the two memset
// nodes of a full-size reset + one-frame push did not order against the kernels behind them on ROCm 7.2 -- the first replay was right, later replays read the
// previous replay's state (tools/graph_reset_probe.py: 24576 / 131071 state words off; a reset-only graph and the 16-frame graph were fine).  Kernel nodes are.
__global__ void zero_states_kernel(u32x4* __restrict__ a, size_t na, u32x4* __restrict__ b, size_t nb) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < na) a[t] = u32x4{0, 0, 0, 0};
    else if (t - na < nb) b[t - na] = u32x4{0, 0, 0, 0};
}
extern "C" int sm_stream_reset(sm_stream* s, void* stream) {
    SM_REQUIRE(s, "sm_stream_reset: null");
    { int jrc = auto_join(s, stream); if (jrc) return jrc; }
    hipStream_t st = (hipStream_t)stream;
    {   // DevBuf allocations are 256-byte aligned and the state sizes are multiples of 16 bytes (d_inner * d_conv / d_state fp32 words, d_inner % 4 == 0)
        const size_t na = s->conv_state.bytes / 16, nb = s->ssm_state.bytes / 16;      // none in a model without connector
        SM_REQUIRE(s->conv_state.bytes % 16 == 0 && s->ssm_state.bytes % 16 == 0, "sm_stream_reset: state sizes");
        if (na + nb) {
            zero_states_kernel<<<(unsigned)((na + nb + 255) / 256), 256, 0, st>>>((u32x4*)s->conv_state.p, na, (u32x4*)s->ssm_state.p, nb);
            SM_LAUNCH_CHECK();
        }
    }
    s->T = 0; s->kv_len = 0;
    return SM_OK;
}
extern "C" void sm_stream_close(sm_stream* s) { delete s; }
extern "C" int sm_stream_num_frames(sm_stream* s) { return s ? s->T : -1; }
extern "C" const float* sm_stream_tokens(sm_stream* s) { return s ? s->tokens.as<float>() : nullptr; }
extern "C" int sm_stream_kv_len(sm_stream* s) { return s ? s->kv_len : -1; }
extern "C" int sm_stream_set_kv_len(sm_stream* s, int n) {
    SM_REQUIRE(s && n >= 0 && n <= s->kv_len, "sm_stream_set_kv_len: %d outside [0, kv_len]", n);
    s->kv_len = n;
    return SM_OK;
}
extern "C" const float* sm_stream_logits(sm_stream* s) { return s ? s->lmlog.as<float>() : nullptr; }
extern "C" int sm_stream_read_tokens(sm_stream* s, int t0, int n, float* out, void* stream) {
    SM_REQUIRE(s && out && t0 >= 0 && n >= 0 && t0 + n <= s->T, "sm_stream_read_tokens: [%d, %d) outside [0, %d)", t0, t0 + n, s ? s->T : 0);
    { int jrc = auto_join(s, stream); if (jrc) return jrc; }
    if (n == 0) return SM_OK;
    const int d = s->m->c.conn_d_model;
    SM_HIP(hipMemcpyAsync(out, s->tokens.as<float>() + (size_t)t0 * d, (size_t)n * d * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SM_OK;
}
extern "C" int sm_stream_read_state(sm_stream* s, float* conv_out, float* ssm_out, void* stream) {
    SM_REQUIRE(s && s->conv_state.bytes && s->ssm_state.bytes, "sm_stream_read_state: null stream / model without connector");
    { int jrc = auto_join(s, stream); if (jrc) return jrc; }
    if (conv_out) SM_HIP(hipMemcpyAsync(conv_out, s->conv_state.p, s->conv_state.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (ssm_out) SM_HIP(hipMemcpyAsync(ssm_out, s->ssm_state.p, s->ssm_state.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SM_OK;
}
extern "C" int sm_stream_write_tokens(sm_stream* s, int t0, int n, const float* src, void* stream) {
    SM_REQUIRE(s && src && t0 >= 0 && n > 0 && t0 <= s->T && t0 + n <= s->max_frames, "sm_stream_write_tokens: [%d, %d) not appendable (T=%d, cap=%d)", t0, t0 + n, s ? s->T : 0, s ? s->max_frames : 0);
    { int jrc = auto_join(s, stream); if (jrc) return jrc; }
    const int d = s->m->c.conn_d_model;
    SM_HIP(hipMemcpyAsync(s->tokens.as<float>() + (size_t)t0 * d, src, (size_t)n * d * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (t0 + n > s->T) s->T = t0 + n;
    return SM_OK;
}
extern "C" int sm_stream_read_logits(sm_stream* s, float* out, int32_t* next_token_out, void* stream) {
    SM_REQUIRE(s && s->m->c.llm_layers > 0, "sm_stream_read_logits: perception-only model");
    if (out) SM_HIP(hipMemcpyAsync(out, s->lmlog.p, (size_t)s->m->c.llm_vocab * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (next_token_out) SM_HIP(hipMemcpyAsync(next_token_out, s->next_tok.p, 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SM_OK;
}

// a5-a9 for S streams x F new frames each (rows stream-major, M = S*F <= 32: ONE pass over the connector + gate weights):
// PreNet -> LN -> Mamba step on each stream's own conv / ssm state -> +res -> LN_f -> PostNet -> frame tokens (appended to
// each stream's store) -> 4-layer gate (V/O shortcut) on every token independently -> logits [M][2], decisions [M]
static int conn_gate_pass(sm_model* m, ConnScratch& ws, const float* pooled, int S, int F, const SmSegStates& conv, const SmSegStates& ssm,
                          const SmSegStates& tokdst, float* logits, int32_t* decisions, void* stream) {
    const sm_config_t& c = m->c;
    SM_REQUIRE(c.conn_d_state > 0, "this model was created without the Mamba connector / event gate (conn_d_state == 0): frames cannot be pushed, "
                                   "hand tokens over with sm_stream_write_tokens");
    const int M = S * F;
    const int d = c.conn_d_model, di = c.conn_expand * d, R = c.conn_dt_rank, ds = c.conn_d_state;
    const int xd = cdiv(R + 2 * ds, 32) * 32 + 32;
    const int pr = c.gate_precise;
    const sm_model::Resolved& W = m->R;
    int rc;
    auto L = [&](const Slot* slot, const float* x, int ldx) {
        sm_linear_t a = lin(m, *slot, x, SM_X_F32, M, ldx);
        a.precise = pr;
        return a;
    };
    float* tok = S == 1 ? tokdst.p[0] : ws.tokrows.as<float>();     // one stream: PostNet writes straight into its token store
    // Every norm of the pass rides behind the product that finishes its input row (sm_linear_t.post_ln_*: LayerNorm with beta, RMSNorm
    // without, optional activation, fp32 out): with 17..32 rows the products run as K-slice slabs and the slab sum, epilogue and norm are
    // one launch; with fewer rows the call ends with the norm launch that used to be issued here.
    auto post_norm = [&](sm_linear_t& a, const float* gamma, const float* beta, float eps, int act, float* out) {
        a.post_ln_gamma = gamma; a.post_ln_beta = beta; a.post_ln_eps = eps; a.post_ln_act = act; a.post_ln_out_f32 = out; a.post_ln_ldo = d;
    };
    {   // PreNet: leaky_relu(W x + b)                                         builder.py:166-170
        sm_linear_t a = L(W.pre, pooled, c.conn_mm_hidden);
        a.bias = W.pre_b; a.act = SM_ACT_LEAKY_RELU; a.out_f32 = ws.t0.as<float>(); a.ldo = d;
        // Block: hidden = Mamba(LN(residual)), residual = t0                  block.py:51-67
        post_norm(a, W.cn_w, W.cn_b, c.conn_eps, 0, ws.u.as<float>());
        if ((rc = sm_linear(&a, stream))) return rc;
    }
    {   sm_linear_t a = L(W.in_proj, ws.u.as<float>(), d); a.out_f32 = ws.xz.as<float>(); a.ldo = 2 * di;
        if ((rc = sm_linear(&a, stream))) return rc; }
    if ((rc = sm_mamba_conv_step_seg(ws.xz.as<float>(), S, F, di, c.conn_d_conv, conv, W.conv_w, W.conv_b, ws.xc.as<float>(), stream))) return rc;
    {   sm_linear_t a = L(W.x_proj, ws.xc.as<float>(), di); a.out_f32 = ws.xdbl.as<float>(); a.ldo = xd;
        if ((rc = sm_linear(&a, stream))) return rc; }
    {   // dt = softplus(W_dt dt_r + b): K = dt_rank is padded to 32 inside the packed weight (zeros), x_dbl rows are xd wide
        sm_linear_t a = L(W.dt_proj, ws.xdbl.as<float>(), xd);
        a.bias = W.dt_b; a.act = SM_ACT_SOFTPLUS; a.out_f32 = ws.delta.as<float>(); a.ldo = di;
        if ((rc = sm_linear(&a, stream))) return rc; }
    if ((rc = sm_mamba_ssm_step_seg(ws.xc.as<float>(), ws.delta.as<float>(), ws.xdbl.as<float>(), xd, R, ws.xz.as<float>(), S, F, di, ds, W.A_log, W.Dp, ssm, ws.y.as<float>(), stream))) return rc;
    {   sm_linear_t a = L(W.out_proj, ws.y.as<float>(), di);
        a.residual = ws.t0.as<float>(); a.ldr = d; a.out_f32 = ws.r.as<float>(); a.ldo = d;       // hidden + residual, ssm.py:83
        post_norm(a, W.nf_w, W.nf_b, c.conn_eps, SM_ACT_LEAKY_RELU, ws.lnf.as<float>());           // leaky_relu(norm_f(.)) in front of PostNet
        if ((rc = sm_linear(&a, stream))) return rc; }
    {   sm_linear_t a = L(W.post, ws.lnf.as<float>(), d);
        a.bias = W.post_b; a.out_f32 = tok; a.ldo = d;
        if (c.gate_layers > 0) post_norm(a, W.gate[0].ln1_w, nullptr, c.gate_eps, 0, ws.hn.as<float>());      // the first gate layer's input norm
        else post_norm(a, W.gate_norm, nullptr, c.gate_eps, 0, ws.hfin.as<float>());
        if ((rc = sm_linear(&a, stream))) return rc; }
    if (S > 1 && (rc = sm_scatter_rows(tok, S, F, d, tokdst, stream))) return rc;
    // ---- gate on each of the M tokens independently (seq-len 1 each; builder.py:553-562)
    const int gdh = c.gate_hidden / c.gate_heads, kvn = c.gate_kv_heads * gdh, qn = c.gate_heads * gdh;
    const int rep = c.gate_heads / c.gate_kv_heads;
    // repeat_kv folded into o_proj's operand addressing (sm_linear_t.x_rep) where the kernels read x through it: 16-bit weights, powers of two
    const bool fold_rep = rep > 1 && (rep & (rep - 1)) == 0 && (gdh & (gdh - 1)) == 0 && gdh >= 8;
    const float* hcur = tok;       // layer 0 reads the token, writes ws.h
    static const bool no_tail = [] { const char* e = getenv("SM_GATE_TAIL"); return e && atoi(e) == 0; }();
    const bool tail = m->gate_head_f32.p != nullptr && c.gate_layers > 0 && !no_tail;
    for (int l = 0; l < c.gate_layers; ++l) {
        const sm_model::LayerW& w = W.gate[l];
        {   sm_linear_t a = L(w.v, ws.hn.as<float>(), d); a.out_f32 = ws.v.as<float>(); a.ldo = kvn;
            if ((rc = sm_linear(&a, stream))) return rc; }
        const bool fold = fold_rep && !w.o->fp8;
        if (!fold && (rc = sm_repeat_kv(ws.v.as<float>(), M, c.gate_kv_heads, c.gate_heads, gdh, ws.vrep.as<float>(), stream))) return rc;
        {   sm_linear_t a = fold ? L(w.o, ws.v.as<float>(), kvn) : L(w.o, ws.vrep.as<float>(), qn);
            if (fold) { a.x_rep = rep; a.x_rep_dh = gdh; }
            a.residual = hcur; a.ldr = d; a.out_f32 = ws.h.as<float>(); a.ldo = d;
            post_norm(a, w.ln2_w, nullptr, c.gate_eps, 0, ws.hn.as<float>());
            if ((rc = sm_linear(&a, stream))) return rc; }
        hcur = ws.h.as<float>();
        {   const Slot& gu = *w.gu;
            sm_linear_t a = L(w.gu, ws.hn.as<float>(), d);
            a.N = c.gate_mlp;
            if (gu.fp8) { a.w2 = (const char*)gu.buf.p + (size_t)(c.gate_mlp / 16) * ((((d + 31) / 32) + 1) / 2) * 1024; a.w2_scale = gu.scale.as<float>() + c.gate_mlp; }
            else a.w2 = gu.buf.as<bf16_t>() + (size_t)(c.gate_mlp / 16) * ((d + 31) / 32) * 512;
            a.out_f32 = ws.act.as<float>(); a.ldo = c.gate_mlp;
            if ((rc = sm_linear(&a, stream))) return rc; }
        {   sm_linear_t a = L(w.down, ws.act.as<float>(), c.gate_mlp);
            a.residual = hcur; a.ldr = d; a.out_f32 = ws.h.as<float>(); a.ldo = d;
            if (l + 1 < c.gate_layers) post_norm(a, W.gate[l + 1].ln1_w, nullptr, c.gate_eps, 0, ws.hn.as<float>());
            else if (!tail) post_norm(a, W.gate_norm, nullptr, c.gate_eps, 0, ws.hfin.as<float>());
            if ((rc = sm_linear(&a, stream))) return rc; }
    }
    float* lg = logits ? logits : ws.logits2.as<float>();
    if (tail)       // final RMSNorm + 2-way head + decision in one launch (fp32 throughout)
        return sm_gate_tail(hcur, M, d, d, W.gate_norm, c.gate_eps, m->gate_head_f32.as<float>(), lg, decisions, stream);
    {   sm_linear_t a = L(W.gate_head, ws.hfin.as<float>(), d); a.out_f32 = lg; a.ldo = 2;
        if ((rc = sm_linear(&a, stream))) return rc; }
    if (decisions && (rc = sm_gate_decide(lg, M, decisions, stream))) return rc;
    return SM_OK;
}

extern "C" int sm_stream_push_pooled(sm_stream* s, const float* pooled, int M, float* logits, int32_t* decisions, void* stream) {
    SM_REQUIRE(s && pooled && M >= 1 && M <= 32, "sm_stream_push_pooled: M=%d outside [1,32]", M);
    SM_REQUIRE(s->T + M <= s->max_frames, "sm_stream_push_pooled: token store full (%d + %d > %d)", s->T, M, s->max_frames);
    { int jrc = auto_join(s, stream); if (jrc) return jrc; }
    SmSegStates conv, ssm, tok;
    conv.p[0] = s->conv_state.as<float>(); ssm.p[0] = s->ssm_state.as<float>();
    tok.p[0] = s->tokens.as<float>() + (size_t)s->T * s->m->c.conn_d_model;
    int rc = conn_gate_pass(s->m, s->w, pooled, 1, M, conv, ssm, tok, logits, decisions, stream);
    if (rc) return rc;
    s->T += M;
    return SM_OK;
}

extern "C" int sm_stream_push_frames(sm_stream* s, const uint8_t* frames, int M, float* logits, int32_t* decisions, void* stream) {
    SM_REQUIRE(s && frames && M >= 1 && M <= s->m->Bmax, "sm_stream_push_frames: M=%d outside [1, max_frames_per_call=%d]", M, s ? s->m->Bmax : 0);
    // one ViT batch, then the connector+gate in frame order, at most 32 frames per weight pass
    int rc = sm_vit_encode(s->m, frames, M, s->w.pooled.as<float>(), nullptr, nullptr, stream);
    if (rc) return rc;
    const int cap = 32;       // (fp8 weights too: the 17..32-row weight-streaming kernel reads the fp8 image; the gate is weight-only in both fp8 modes)
    const int parts = cdiv(M, cap), per = cdiv(M, parts);
    for (int i = 0; i < M; i += per) {
        const int n = M - i < per ? M - i : per;
        rc = sm_stream_push_pooled(s, s->w.pooled.as<float>() + (size_t)i * s->m->c.conn_mm_hidden, n, logits ? logits + 2 * i : nullptr,
                                   decisions ? decisions + i : nullptr, stream);
        if (rc) return rc;
    }
    return SM_OK;
}

// Pipelined form of sm_stream_push_frames.  The tower of this call runs on the caller's stream; its connector + gate pass runs on
// the stream's side HIP stream, ordered behind the tower by an event and behind the previous call's pass by stream order.  The
// caller's stream is free to start the next call's tower at once: the weight-streaming pass (memory-bound, ~5 % of a 28-frame
// step) then runs in the launch gaps and tails of the next tower's MFMA kernels.  Results are identical; `logits` / `decisions`
// are complete for the caller's stream after sm_stream_join (every other sm_stream_* call joins by itself).
extern "C" int sm_stream_push_frames_pipelined(sm_stream* s, const uint8_t* frames, int M, float* logits, int32_t* decisions, void* stream) {
    SM_REQUIRE(s && frames && M >= 1 && M <= s->m->Bmax, "sm_stream_push_frames_pipelined: M=%d outside [1, max_frames_per_call=%d]", M, s ? s->m->Bmax : 0);
    SM_REQUIRE(s->T + M <= s->max_frames, "sm_stream_push_frames_pipelined: token store full (%d + %d > %d)", s->T, M, s->max_frames);
    sm_model* m = s->m;
    hipStream_t st = (hipStream_t)stream;
    if (!s->side) {
        SM_HIP(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        SM_HIP(hipEventCreateWithFlags(&s->ev_vit, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) {
            SM_HIP(hipEventCreateWithFlags(&s->ev_pass[i], hipEventDisableTiming));
            int rc = s->pooled_pp[i].alloc((size_t)m->Bmax * m->c.conn_mm_hidden * 4);
            if (rc) return rc;
        }
    }
    const int f = s->flip;
    // pooled_pp[f] was last read by the pass issued two calls ago: the tower must not overwrite it earlier
    SM_HIP(hipStreamWaitEvent(st, s->ev_pass[f], 0));      // (an event that was never recorded counts as complete: no wait)
    float* pooled = s->pooled_pp[f].as<float>();
    int rc = sm_vit_encode(m, frames, M, pooled, nullptr, nullptr, stream);
    if (rc) return rc;
    SM_HIP(hipEventRecord(s->ev_vit, st));
    SM_HIP(hipStreamWaitEvent(s->side, s->ev_vit, 0));
    const int cap = 32;
    const int parts = cdiv(M, cap), per = cdiv(M, parts);
    const int d = m->c.conn_d_model;
    for (int i = 0; i < M; i += per) {
        const int n = M - i < per ? M - i : per;
        SmSegStates conv, ssm, tok;
        conv.p[0] = s->conv_state.as<float>(); ssm.p[0] = s->ssm_state.as<float>();
        tok.p[0] = s->tokens.as<float>() + (size_t)s->T * d;
        rc = conn_gate_pass(m, s->w, pooled + (size_t)i * m->c.conn_mm_hidden, 1, n, conv, ssm, tok, logits ? logits + 2 * i : nullptr,
                            decisions ? decisions + i : nullptr, (void*)s->side);
        if (rc) return rc;
        s->T += n;
    }
    SM_HIP(hipEventRecord(s->ev_pass[f], s->side));
    s->last = f;
    s->pass_seq += 1;
    s->flip ^= 1;
    return SM_OK;
}

extern "C" int sm_stream_join(sm_stream* s, void* stream) {
    SM_REQUIRE(s, "sm_stream_join: null");
    return auto_join(s, stream);
}
extern "C" int sm_stream_pass_ticket(sm_stream* s) { return (s && s->side) ? (s->flip ^ 1) : -1; }
extern "C" int sm_stream_join_ticket(sm_stream* s, int ticket, void* stream) {
    SM_REQUIRE(s && (ticket == 0 || ticket == 1) && s->side, "sm_stream_join_ticket: bad ticket %d / no pipelined call yet", ticket);
    SM_HIP(hipStreamWaitEvent((hipStream_t)stream, s->ev_pass[ticket], 0));
    return SM_OK;
}

// ---- stream group: one tick of S streams through ONE ViT batch and ONE connector + gate weight pass per <= 32 rows.
// The reference serves one stream per model object, one frame per call (eval/video_score_stream_demo.py:283-299; "only
// support batch size 1", eval/inference_video_score_stream_ddp.py:325): at one frame per call the ViT GEMMs run at M = 577 and
// the gate streams 1.6 GB of weights per frame.  Batching ACROSS streams keeps the per-frame decision latency of one frame
// while the kernels see S x 577 rows -- results per stream are those of S independent one-frame calls.
struct sm_stream_group {
    sm_model* m;
    std::vector<sm_stream*> streams;
    ConnScratch w;
    // batched decode scratch (rows = streams), allocated by the first sm_group_llm_decode
    DevBuf d_emb, d_xnb, d_xn32, d_qkvf, d_qb, d_ctxb, d_actb, d_log, d_ws;
    DevBuf d_tab;                // > SM_MAX_SEG streams: [layers][S] K pointers, [layers][S] V^T pointers, [S] start positions (SmDecodeSegTab), uploaded once per call
    bool d_ready = false;
};

extern "C" int sm_group_create(sm_stream** streams, int S, sm_stream_group** out) {
    SM_REQUIRE(streams && out && S >= 1, "sm_group_create: bad args");
    for (int i = 0; i < S; ++i) {
        SM_REQUIRE(streams[i] && streams[i]->m == streams[0]->m, "sm_group_create: stream %d is null or belongs to another model", i);
        for (int j = 0; j < i; ++j) SM_REQUIRE(streams[i] != streams[j], "sm_group_create: stream %d listed twice", i);
    }
    sm_stream_group* g = new sm_stream_group();
    g->m = streams[0]->m;
    g->streams.assign(streams, streams + S);
    int rc = g->w.alloc(g->m);
    if (rc) { delete g; return rc; }
    *out = g;
    return SM_OK;
}
extern "C" void sm_group_destroy(sm_stream_group* g) { delete g; }
extern "C" int sm_group_size(sm_stream_group* g) { return g ? (int)g->streams.size() : -1; }

// pooled [S][F][mm_hidden] (stream-major) -> logits [S][F][2], decisions [S][F]
extern "C" int sm_group_push_pooled(sm_stream_group* g, const float* pooled, int F, float* logits, int32_t* decisions, void* stream) {
    SM_REQUIRE(g && pooled && F >= 1, "sm_group_push_pooled: bad args");
    sm_model* m = g->m;
    const int S = (int)g->streams.size(), d = m->c.conn_d_model;
    const int cap = 32;
    SM_REQUIRE(F <= cap, "sm_group_push_pooled: %d frames per stream exceed one weight pass (%d rows)", F, cap);
    for (int i = 0; i < S; ++i)
        SM_REQUIRE(g->streams[i]->T + F <= g->streams[i]->max_frames, "sm_group_push_pooled: token store of stream %d full (%d + %d > %d)", i,
                   g->streams[i]->T, F, g->streams[i]->max_frames);
    for (int i = 0; i < S; ++i) { int jrc = auto_join(g->streams[i], stream); if (jrc) return jrc; }
    const int per = cap / F;                       // whole streams per weight pass
    for (int s0 = 0; s0 < S; s0 += per) {
        const int n = S - s0 < per ? S - s0 : per;
        SmSegStates conv, ssm, tok;
        for (int i = 0; i < n; ++i) {
            sm_stream* st = g->streams[s0 + i];
            conv.p[i] = st->conv_state.as<float>(); ssm.p[i] = st->ssm_state.as<float>();
            tok.p[i] = st->tokens.as<float>() + (size_t)st->T * d;
        }
        int rc = conn_gate_pass(m, g->w, pooled + (size_t)s0 * F * m->c.conn_mm_hidden, n, F, conv, ssm, tok,
                                logits ? logits + (size_t)2 * s0 * F : nullptr, decisions ? decisions + (size_t)s0 * F : nullptr, stream);
        if (rc) return rc;
        for (int i = 0; i < n; ++i) g->streams[s0 + i]->T += F;
    }
    return SM_OK;
}

// frames u8 [S][F][H][W][3] (stream-major: the F new frames of stream 0, then of stream 1, ...)
extern "C" int sm_group_push_frames(sm_stream_group* g, const uint8_t* frames, int F, float* logits, int32_t* decisions, void* stream) {
    SM_REQUIRE(g && frames && F >= 1, "sm_group_push_frames: bad args");
    const int S = (int)g->streams.size();
    SM_REQUIRE(S * F <= g->m->Bmax, "sm_group_push_frames: %d streams x %d frames exceed max_frames_per_call=%d", S, F, g->m->Bmax);
    int rc = sm_vit_encode(g->m, frames, S * F, g->w.pooled.as<float>(), nullptr, nullptr, stream);
    if (rc) return rc;
    return sm_group_push_pooled(g, g->w.pooled.as<float>(), F, logits, decisions, stream);
}

// ------------------------------------------------------------------------------------------------ LLM
// SM_NO_FUSED_NORM=1: keep the separate RMSNorm launches on the decode path (A/B tuning switch)
static const bool g_no_fused_norm = [] { const char* e = getenv("SM_NO_FUSED_NORM"); return e && atoi(e) != 0; }();
// SM_NO_FUSED_ROPE=1: separate rope_kv_kernel launch behind the q/k/v product of a decode step (A/B tuning switch)
static const bool g_no_fused_rope = [] { const char* e = getenv("SM_NO_FUSED_ROPE"); return e && atoi(e) != 0; }();

// one pass of the decoder over n rows of W->emb (fp32 residual stream) at positions kv_len..kv_len+n-1
static int llm_layers(sm_stream* s, sm_model::LlmWs* W, int n, void* stream) {
    sm_model* m = s->m;
    const sm_config_t& c = m->c;
    const int ld = c.llm_hidden, H = c.llm_heads, KV = c.llm_kv_heads, dh = ld / H, qn = H * dh, kn = KV * dh;
    float* x = W->emb.as<float>();
    int rc;
    const int f16 = c.llm_fp16 ? 1 : 0, od = f16 ? SM_OP_F16 : SM_OP_BF16;      // 16-bit type of every LLM operand / cache (weights are packed to match)
    for (int l = 0; l < c.llm_layers; ++l) {
        const sm_model::LayerW& w = m->R.llm[l];
        // decode (one row): both RMSNorms ride inside the weight-streaming products that consume them
        const bool fuse_norm = n == 1 && (ld & 31) == 0 && !g_no_fused_norm;
        // chunks of rows (prefill, teacher-forced evaluation): the RMSNorms ride BEHIND the residual products (sm_linear_t.post_ln_*: o_proj leaves
        // ln2(x), down_proj the next layer's ln1(x) as the 16-bit operand W->xnb) -- one pass with the slab sum where the product runs as split-K
        // slabs (2048 rows: the 256 x 256 kernel's N = 4096 shapes), the same sm_norm_ex launch as before everywhere else.  Only layer 0's ln1 is a
        // call of its own.
        if (!fuse_norm && l == 0 && (rc = sm_norm_ex(x, n, ld, ld, w.ln1_w, nullptr, c.llm_eps, 0, nullptr, W->xnb.p, ld, od, stream))) return rc;
        // decode: RoPE + KV append ride in the epilogue of the q/k/v product (no fp32 q/k/v round trip, one launch less per layer)
        const bool fuse_rope = fuse_norm && dh == 128 && ld >= 1024 && !g_no_fused_rope;
        {   sm_linear_t a = fuse_norm ? lin(m, *w.qkv, x, SM_X_F32, n, ld) : lin(m, *w.qkv, W->xnb.p, SM_X_BF16, n, ld);
            if (fuse_norm) { a.norm_gamma = w.ln1_w; a.norm_eps = c.llm_eps; }
            if (fuse_rope) {
                SmRopeEpi re;
                re.cos_tab = m->rope_cos.as<float>(); re.sin_tab = m->rope_sin.as<float>(); re.q = W->qb.p;
                re.H = H; re.KV = KV; re.S_max = s->cap;
                re.seg.kc[0] = s->kc[l].p; re.seg.vtc[0] = s->vtc[l].p; re.seg.pos[0] = s->kv_len;
                if ((rc = sm_linear_qkv_rope(&a, re, stream))) return rc;
            } else {
                a.out_f32 = W->qkvf.as<float>(); a.ldo = qn + 2 * kn;
                if ((rc = sm_linear(&a, stream))) return rc;
            }
        }
        if (!fuse_rope && (rc = sm_rope_kv_append_ex(W->qkvf.as<float>(), n, s->kv_len, H, KV, dh, m->rope_cos.as<float>(), m->rope_sin.as<float>(), W->qb.p, s->kc[l].p, s->vtc[l].p, s->cap, f16, stream))) return rc;
        if (n == 1) {
            if ((rc = sm_llm_decode_attention_ex(W->qb.p, s->kc[l].p, s->vtc[l].p, s->kv_len, H, KV, dh, s->cap, W->attn_ws.as<float>(), SM_DECODE_SPLITS, W->ctxb.p, f16, stream, c.llm_sliding_window))) return rc;
        } else if ((rc = sm_llm_attention_ex(W->qb.p, s->kc[l].p, s->vtc[l].p, n, s->kv_len, H, KV, dh, s->cap, W->ctxb.p, f16, stream, c.llm_sliding_window))) return rc;
        {   sm_linear_t a = lin(m, *w.o, W->ctxb.p, SM_X_BF16, n, qn);
            a.residual = x; a.ldr = ld; a.out_f32 = x; a.ldo = ld;
            if (!fuse_norm) { a.post_ln_gamma = w.ln2_w; a.post_ln_eps = c.llm_eps; a.post_ln_out = W->xnb.p; a.post_ln_ldo = ld; }
            if ((rc = sm_linear(&a, stream))) return rc; }
        if (n <= (c.weights_fp8 == 2 ? 16 : 32)) {   // decode / tiny chunks: SwiGLU fused into the dual weight-streaming kernel (fp8 MFMA mode: above 16 rows the tiled fp8 product)
            const Slot& gu = *w.gu;
            sm_linear_t a = fuse_norm ? lin(m, gu, x, SM_X_F32, n, ld) : lin(m, gu, W->xnb.p, SM_X_BF16, n, ld);
            if (fuse_norm) { a.norm_gamma = w.ln2_w; a.norm_eps = c.llm_eps; }
            a.N = c.llm_mlp;
            if (gu.fp8) { a.w2 = (const char*)gu.buf.p + (size_t)(c.llm_mlp / 16) * ((ld / 32 + 1) / 2) * 1024; a.w2_scale = gu.scale.as<float>() + c.llm_mlp; }
            else a.w2 = gu.buf.as<bf16_t>() + (size_t)(c.llm_mlp / 16) * (ld / 32) * 512;
            a.out_bf16 = W->actb.p; a.ldo_bf16 = c.llm_mlp;
            if ((rc = sm_linear(&a, stream))) return rc;
        } else {
            // prefill chunks / teacher-forced rows: act_fn(gate) * up in the epilogue of the gate | up product where the 256 x 256 kernel runs it (>= 192 tiles:
            // gate and up fragments of a column meet in one lane, 16-bit rows out: no fp32 [n][2 mlp] round trip, no SwiGLU launch); below that
            // sm_linear runs the product into its own fp32 scratch + the SwiGLU pass, as before
            sm_linear_t a = lin(m, *w.gu, W->xnb.p, SM_X_BF16, n, ld);
            a.act = SM_ACT_SWIGLU_DUAL;
            a.out_bf16 = W->actb.p; a.ldo_bf16 = c.llm_mlp;
            if ((rc = sm_linear(&a, stream))) return rc;
        }
        {   sm_linear_t a = lin(m, *w.down, W->actb.p, SM_X_BF16, n, c.llm_mlp);
            a.residual = x; a.ldr = ld; a.out_f32 = x; a.ldo = ld;
            if (!fuse_norm && l + 1 < c.llm_layers) { a.post_ln_gamma = m->R.llm[l + 1].ln1_w; a.post_ln_eps = c.llm_eps; a.post_ln_out = W->xnb.p; a.post_ln_ldo = ld; }
            if ((rc = sm_linear(&a, stream))) return rc; }
    }
    s->kv_len += n;
    return SM_OK;
}

// final norm + lm_head on row `row` of the residual stream, greedy argmax into next_tok; decode steps pass where the NEXT step's
// output id goes and ask for its embedding row in W->emb (row 0) -- nullptr / false after a prefill and on the last step
static int llm_head(sm_stream* s, sm_model::LlmWs* W, int row, void* stream, int32_t* emit_next = nullptr, bool embed_next = false) {
    sm_model* m = s->m;
    const sm_config_t& c = m->c;
    const int ld = c.llm_hidden;
    int rc;
    const bool fuse_norm = (ld & 31) == 0 && !g_no_fused_norm;
    const int f16 = c.llm_fp16 ? 1 : 0, od = f16 ? SM_OP_F16 : SM_OP_BF16;
    if (!fuse_norm && (rc = sm_norm_ex(W->emb.as<float>() + (size_t)row * ld, 1, ld, ld, m->R.llm_norm, nullptr, c.llm_eps, 0, nullptr, W->xnb.p, ld, od, stream))) return rc;
    sm_linear_t a = fuse_norm ? lin(m, *m->R.lm_head, W->emb.as<float>() + (size_t)row * ld, SM_X_F32, 1, ld)
                              : lin(m, *m->R.lm_head, W->xnb.p, SM_X_BF16, 1, ld);
    if (fuse_norm) { a.norm_gamma = m->R.llm_norm; a.norm_eps = c.llm_eps; }
    a.out_f32 = s->lmlog.as<float>(); a.ldo = c.llm_vocab;
    if ((rc = sm_linear(&a, stream))) return rc;
    if (!emit_next && !embed_next) return sm_argmax(s->lmlog.as<float>(), c.llm_vocab, s->next_tok.as<int32_t>(), stream);
    argmax_emit_embed_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(s->lmlog.as<float>(), c.llm_vocab, s->next_tok.as<int32_t>(), emit_next,
                                                                 m->R.embed->buf.as<bf16_t>(), ld, embed_next ? W->emb.as<float>() : nullptr, f16);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

extern "C" int sm_llm_prefill(sm_stream* s, const int32_t* ids, int n, void* stream) {
    SM_REQUIRE(s && ids && n > 0 && s->m->c.llm_layers > 0, "sm_llm_prefill: bad args / perception-only model");
    SM_REQUIRE(s->kv_len + n <= s->max_seq, "sm_llm_prefill: context %d + %d exceeds max_seq %d", s->kv_len, n, s->max_seq);
    { int jrc = auto_join(s, stream); if (jrc) return jrc; }
    sm_model* m = s->m;
    const int ld = m->c.llm_hidden;
    int rc, done = 0, last_rows = 0;
    sm_model::LlmWs* W;
    if ((rc = m->llm_workspace(stream, &W)) || (rc = kv_reserve(s, s->kv_len + n, stream))) return rc;
    while (done < n) {
        int cur = n - done < sm_model::LLM_CHUNK ? n - done : sm_model::LLM_CHUNK;
        if ((rc = sm_embed_splice_ex(ids + done, cur, m->R.embed->buf.p, s->tokens.as<float>(), ld, W->emb.as<float>(), m->c.llm_fp16 ? 1 : 0, m->c.llm_vocab, s->max_frames, stream))) return rc;
        if ((rc = llm_layers(s, W, cur, stream))) return rc;
        done += cur; last_rows = cur;
    }
    return llm_head(s, W, last_rows - 1, stream);
}

extern "C" int sm_llm_forward_logits(sm_stream* s, const int32_t* ids, int n, float* logits, void* stream) {
    SM_REQUIRE(s && ids && logits && n > 0 && s->m->c.llm_layers > 0, "sm_llm_forward_logits: bad args / perception-only model");
    SM_REQUIRE(s->kv_len + n <= s->max_seq, "sm_llm_forward_logits: context %d + %d exceeds max_seq %d", s->kv_len, n, s->max_seq);
    { int jrc = auto_join(s, stream); if (jrc) return jrc; }
    sm_model* m = s->m;
    const sm_config_t& c = m->c;
    const int ld = c.llm_hidden, V = c.llm_vocab;
    int rc, done = 0;
    sm_model::LlmWs* W;
    if ((rc = m->llm_workspace(stream, &W)) || (rc = kv_reserve(s, s->kv_len + n, stream))) return rc;
    while (done < n) {
        const int cur = n - done < sm_model::LLM_CHUNK ? n - done : sm_model::LLM_CHUNK;
        if ((rc = sm_embed_splice_ex(ids + done, cur, m->R.embed->buf.p, s->tokens.as<float>(), ld, W->emb.as<float>(), m->c.llm_fp16 ? 1 : 0, m->c.llm_vocab, s->max_frames, stream))) return rc;
        if ((rc = llm_layers(s, W, cur, stream))) return rc;
        // final norm + lm_head on all `cur` rows of this chunk (llm_head does the last row only)
        if ((rc = sm_norm_ex(W->emb.as<float>(), cur, ld, ld, m->R.llm_norm, nullptr, c.llm_eps, 0, nullptr, W->xnb.p, ld, c.llm_fp16 ? SM_OP_F16 : SM_OP_BF16, stream))) return rc;
        sm_linear_t a = lin(m, *m->R.lm_head, W->xnb.p, SM_X_BF16, cur, ld);
        a.out_f32 = logits + (size_t)done * V; a.ldo = V;
        if ((rc = sm_linear(&a, stream))) return rc;
        done += cur;
    }
    // keep the stream's "pending token / last logits" state what a prefill would have left
    SM_HIP(hipMemcpyAsync(s->lmlog.p, logits + (size_t)(n - 1) * V, (size_t)V * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return sm_argmax(s->lmlog.as<float>(), V, s->next_tok.as<int32_t>(), stream);
}

extern "C" int sm_stream_set_next_token(sm_stream* s, const int32_t* tok_dev, void* stream) {
    SM_REQUIRE(s && tok_dev && s->m->c.llm_layers > 0, "sm_stream_set_next_token: bad args / perception-only model");
    copy_i32_kernel<<<1, 1, 0, (hipStream_t)stream>>>(tok_dev, s->next_tok.as<int32_t>());
    SM_LAUNCH_CHECK();
    return SM_OK;
}

extern "C" int sm_llm_decode(sm_stream* s, int n_steps, int32_t* out_ids, void* stream) {
    SM_REQUIRE(s && out_ids && n_steps > 0 && s->m->c.llm_layers > 0, "sm_llm_decode: bad args");
    SM_REQUIRE(s->kv_len + n_steps <= s->max_seq, "sm_llm_decode: context %d + %d exceeds max_seq %d", s->kv_len, n_steps, s->max_seq);
    sm_model* m = s->m;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    sm_model::LlmWs* W;
    if ((rc = m->llm_workspace(stream, &W)) || (rc = kv_reserve(s, s->kv_len + n_steps, stream))) return rc;
    // emit the pending greedy token and feed it back (its KV is appended, the next token becomes pending); from the second step on
    // the emit + embedding gather ride in the previous step's argmax launch
    copy_i32_kernel<<<1, 1, 0, st>>>(s->next_tok.as<int32_t>(), out_ids);
    embed_last_kernel<<<1, 256, 0, st>>>(s->next_tok.as<int32_t>(), m->R.embed->buf.as<bf16_t>(), m->c.llm_hidden, W->emb.as<float>(), m->c.llm_fp16 ? 1 : 0);
    SM_LAUNCH_CHECK();
    for (int j = 0; j < n_steps; ++j) {
        const bool more = j + 1 < n_steps;
        if ((rc = llm_layers(s, W, 1, stream))) return rc;
        if ((rc = llm_head(s, W, 0, stream, more ? out_ids + j + 1 : nullptr, more))) return rc;
    }
    return SM_OK;
}

// ------------------------------------------------------------------------------------------------ batched decode
// n_steps greedy steps of the ACTIVE streams of a group at once: the weight-streaming products run with one row per stream, so
// the 14.2 GB of Mistral-7B weights are read once per step for all of them (batch-1 decode is HBM-bound: S streams decode at
// nearly S times the aggregate tokens/s until the GEMVs turn compute-bound).  Each stream keeps its own KV cache, position and
// pending token; RoPE / append, the flash-decoding attention + merge, the token gather and the arg-max run once per step over
// all streams (per-stream pointers by value).  The reference decodes one stream per model object (HF generate, batch 1:
// videollama2_mistral.py:426-431); per stream the arithmetic is that of sm_llm_decode (same bf16 rounding points; the RMSNorm
// runs as its own launch here, and the fp32 summation order of the skinny products depends on the row count).
extern "C" int sm_group_llm_decode(sm_stream_group* g, const int32_t* active_host, int n_steps, int32_t* out_ids, void* stream) {
    SM_REQUIRE(g && out_ids && n_steps > 0 && g->m->c.llm_layers > 0, "sm_group_llm_decode: bad args / perception-only model");
    sm_model* m = g->m;
    const sm_config_t& c = m->c;
    std::vector<sm_stream*> act;
    std::vector<int> idx;
    for (size_t i = 0; i < g->streams.size(); ++i)
        if (!active_host || active_host[i]) { act.push_back(g->streams[i]); idx.push_back((int)i); }
    const int S = (int)act.size();
    SM_REQUIRE(S >= 1, "sm_group_llm_decode: no active stream");
    // up to SM_MAX_SEG streams: weight-streaming kernels (one row per stream); beyond that, up to SM_GROUP_DECODE_MAX: the same step with the
    // linears on the tiled MFMA GEMM (M = streams) -- still ONE pass over the weights per step -- and the per-stream kernels (token gather,
    // RoPE + KV append, attention, arg-max: per-stream pointers travel by value, SM_MAX_SEG at a time) in chunks
    // (fp8 weights: the weight-streaming kernels read the fp8 image up to 64 rows; beyond that a product would expand it to bf16 per call)
    const int s_cap = c.weights_fp8 == 2 ? 16 : c.weights_fp8 ? (sm_skinny_lds64_on() ? 64 : SM_MAX_SEG) : SM_GROUP_DECODE_MAX;
    SM_REQUIRE(S <= s_cap, "sm_group_llm_decode: %d active streams exceed one weight pass (%d)", S, s_cap);
    const int ld = c.llm_hidden, H = c.llm_heads, KV = c.llm_kv_heads, dh = ld / H, qn = H * dh, kn = KV * dh, V = c.llm_vocab;
    // the per-stream kernels of a step take ONE cache pitch (V^T row stride) for all their streams: the active streams are brought to a common capacity --
    // the largest any of them has or needs for these steps (kv_reserve's geometric rule) -- before the first launch
    int S_max = 0, min_max_seq = act[0]->max_seq;
    for (sm_stream* s : act) {
        SM_REQUIRE(s->kv_len >= 1 && s->kv_len + n_steps <= s->max_seq, "sm_group_llm_decode: a stream has no context or would exceed max_seq (%d + %d > %d)", s->kv_len, n_steps, s->max_seq);
        int jrc = auto_join(s, stream); if (jrc) return jrc;
        int need = s->cap;
        while (need < s->kv_len + n_steps) need *= 2;
        need = (need + 255) / 256 * 256;
        if (need > s->max_seq) need = s->max_seq;
        if (need > S_max) S_max = need;
        if (s->max_seq < min_max_seq) min_max_seq = s->max_seq;
    }
    SM_REQUIRE(S_max <= min_max_seq, "sm_group_llm_decode: the streams of a batched decode share one cache capacity (%d tokens needed) and one of them was opened with max_seq = %d", S_max, min_max_seq);
    if (S > 1)
        for (sm_stream* s : act) { int grc = kv_set_cap(s, S_max, stream); if (grc) return grc; }
    // ONE active stream (a lone reply in a multi-stream session): its own decode loop -- RMSNorm and RoPE ride in the products' kernels there (3.16 -> 2.86 ms per step)
    if (S == 1) return sm_llm_decode(act[0], n_steps, out_ids + (size_t)idx[0] * n_steps, stream);
    int rc = 0;
    if (!g->d_ready) {
        const size_t R = SM_GROUP_DECODE_MAX;
#define A(buf, bytes) if (!rc) rc = g->buf.alloc(bytes)
        A(d_emb, R * ld * 4); A(d_xnb, R * ld * 2); A(d_xn32, (size_t)16 * ld * 4); A(d_qkvf, R * (qn + 2 * kn) * 4); A(d_qb, R * qn * 2); A(d_ctxb, R * qn * 2);
        A(d_actb, R * c.llm_mlp * 2); A(d_log, R * V * 4); A(d_ws, (size_t)SM_MAX_SEG * SM_DECODE_SPLITS * H * (dh + 2) * 4);
#undef A
        if (rc) return rc;
        g->d_ready = true;
    }
    float* x = g->d_emb.as<float>();
    const int f16 = c.llm_fp16 ? 1 : 0, od = f16 ? SM_OP_F16 : SM_OP_BF16;
    const int NC = cdiv(S, SM_MAX_SEG);                       // chunks of the per-stream kernels
    // > SM_MAX_SEG streams: the per-stream cache pointers of every layer and the start positions go to the device ONCE per call (SmDecodeSegTab; 256 KB at 512
    // streams x 32 layers), so that RoPE + append and the attention of ALL streams are one launch each per layer -- they were one launch per 128 streams
    // (pointer packs by value: 4 + 4 launches per layer at 512 streams).  SM_DECODE_TAB=0: the packs (A/B)
    static const bool tab_env = [] { const char* e = getenv("SM_DECODE_TAB"); return !e || atoi(e) != 0; }();
    const bool use_tab = tab_env && NC > 1;
    void* const* tab_kc = nullptr; void* const* tab_vtc = nullptr; const int* tab_pos0 = nullptr;
    int kv_max0 = 0;
    if (use_tab) {
        const size_t np = (size_t)c.llm_layers * S;
        std::vector<void*> hp(2 * np);
        std::vector<int> hpos(S);
        for (int l = 0; l < c.llm_layers; ++l)
            for (int t = 0; t < S; ++t) { hp[(size_t)l * S + t] = act[t]->kc[l].p; hp[np + (size_t)l * S + t] = act[t]->vtc[l].p; }
        for (int t = 0; t < S; ++t) { hpos[t] = act[t]->kv_len; if (act[t]->kv_len > kv_max0) kv_max0 = act[t]->kv_len; }
        const size_t bytes = 2 * np * sizeof(void*) + (size_t)S * sizeof(int);
        if (g->d_tab.bytes < bytes) { SM_HIP(hipStreamSynchronize((hipStream_t)stream)); if ((rc = g->d_tab.alloc(bytes))) return rc; }
        SM_HIP(hipMemcpyAsync(g->d_tab.p, hp.data(), 2 * np * sizeof(void*), hipMemcpyHostToDevice, (hipStream_t)stream));
        SM_HIP(hipMemcpyAsync((char*)g->d_tab.p + 2 * np * sizeof(void*), hpos.data(), (size_t)S * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream));
        SM_HIP(hipStreamSynchronize((hipStream_t)stream));            // the host vectors go out of scope; once per call, not per step
        tab_kc = (void* const*)g->d_tab.p; tab_vtc = tab_kc + np; tab_pos0 = (const int*)((char*)g->d_tab.p + 2 * np * sizeof(void*));
    }
    auto c0 = [&](int ch) { return ch * SM_MAX_SEG; };
    auto cn = [&](int ch) { return S - c0(ch) < SM_MAX_SEG ? S - c0(ch) : SM_MAX_SEG; };
    for (int j = 0; j < n_steps; ++j) {
        // emit the pending token of every stream (out_ids[stream][j], rows of inactive streams untouched) and feed it back
        if (NC > 1) {               // packs of SM_BIG_SEG streams per launch
            for (int b0 = 0; b0 < S; b0 += SM_BIG_SEG) {
                const int bn = S - b0 < SM_BIG_SEG ? S - b0 : SM_BIG_SEG;
                SmTokPtrsBig tok, rows;
                for (int t = 0; t < bn; ++t) { tok.p[t] = act[b0 + t]->next_tok.as<int32_t>(); rows.p[t] = out_ids + (size_t)idx[b0 + t] * n_steps; }
                if ((rc = sm_embed_tokens_seg_big(tok, bn, m->R.embed->buf.p, ld, x + (size_t)b0 * ld, rows, j, f16, stream))) return rc;
            }
        } else
        for (int ch = 0; ch < NC; ++ch) {
            SmTokPtrs tok, rows;
            for (int t = 0; t < cn(ch); ++t) { tok.p[t] = act[c0(ch) + t]->next_tok.as<int32_t>(); rows.p[t] = out_ids + (size_t)idx[c0(ch) + t] * n_steps; }
            if ((rc = sm_embed_tokens_seg(tok, cn(ch), m->R.embed->buf.p, ld, x + (size_t)c0(ch) * ld, rows, j, f16, stream))) return rc;
        }
        // RMSNorms ride behind the products that finish their rows (sm_linear_t.post_ln_*): with 17..32 active streams o_proj and down_proj
        // run as K-slice slabs and the slab sum + residual + the NEXT norm are one launch; otherwise the call ends with the norm launch
        // that used to be issued here.  Only the first layer's input norm is a launch of its own.
        // 2..3 active streams: the RMSNorms ride INSIDE the products that consume them, as in a stream's own decode loop (sm_linear_t.norm_gamma: the
        // normalised rows live in LDS, M x K <= 16384; every block normalises all rows, so the saving ends where that work outgrows two launches --
        // same box: 2 / 3 / 4 streams 3.03 / 3.16 / 3.27 ms per step against 3.17 / 3.20 / 3.23 with the norm launches)
        const bool fuse_norm = S <= 3 && (long)S * ld <= 16384 && (ld & 31) == 0 && !g_no_fused_norm;
        // up to 8 active streams: RoPE + KV append ride in the epilogue of the q/k/v product, every row against ITS stream's cache and position (the per-row
        // form of the kernel a stream's own decode step uses: one launch and the fp32 q/k/v round trip less per layer); its activations are fp32 rows, so the
        // norm in front of it leaves fp32.  Worth 0.4-1 % on bf16 weights and 2-3 % on fp8 weights at 2..8 streams (same box), nothing at 12.
        const bool fuse_rope = S <= 8 && dh == 128 && ld >= 1024 && (ld & 31) == 0 && !g_no_fused_rope;
        if (!fuse_norm && c.llm_layers > 0 &&
            (rc = sm_norm_ex(x, S, ld, ld, m->R.llm[0].ln1_w, nullptr, c.llm_eps, 0, fuse_rope ? g->d_xn32.as<float>() : nullptr, fuse_rope ? nullptr : g->d_xnb.p, ld, od, stream))) return rc;
        for (int l = 0; l < c.llm_layers; ++l) {
            const sm_model::LayerW& w = m->R.llm[l];
            bool rope_done = false, attn_done = false;
            SmSlabOut qkv_slabs = {nullptr, 0, 0};
            {   sm_linear_t a = fuse_norm ? lin(m, *w.qkv, x, SM_X_F32, S, ld) : fuse_rope ? lin(m, *w.qkv, g->d_xn32.p, SM_X_F32, S, ld) : lin(m, *w.qkv, g->d_xnb.p, SM_X_BF16, S, ld);
                if (fuse_norm) { a.norm_gamma = w.ln1_w; a.norm_eps = c.llm_eps; }
                if (fuse_rope) {
                    SmRopeEpi re;
                    re.cos_tab = m->rope_cos.as<float>(); re.sin_tab = m->rope_sin.as<float>(); re.q = g->d_qb.p;
                    re.H = H; re.KV = KV; re.S_max = S_max;
                    for (int t = 0; t < S; ++t) { re.seg.kc[t] = act[t]->kc[l].p; re.seg.vtc[t] = act[t]->vtc[l].p; re.seg.pos[t] = act[t]->kv_len; }
                    if ((rc = sm_linear_qkv_rope(&a, re, stream))) return rc;
                    rope_done = true;
                } else {
                    a.out_f32 = g->d_qkvf.as<float>(); a.ldo = qn + 2 * kn;
                    // with the table path the RoPE kernel can sum the product's split-K slabs itself (33..128 streams: five slabs; one launch less per layer)
                    if (use_tab) { if ((rc = sm_linear_leave_slabs(&a, &qkv_slabs, stream))) return rc; }
                    else if ((rc = sm_linear(&a, stream))) return rc;
                }
            }
            if (NC > 1 && use_tab) {           // every stream in ONE RoPE + append launch and ONE attention launch (device-side pointer table)
                SmDecodeSegTab tab = {tab_kc + (size_t)l * S, tab_vtc + (size_t)l * S, tab_pos0, j};
                const float* qsrc = qkv_slabs.S > 0 ? qkv_slabs.ws : g->d_qkvf.as<float>();
                if ((rc = sm_rope_kv_append_seg_tab(qsrc, S, H, KV, dh, m->rope_cos.as<float>(), m->rope_sin.as<float>(), g->d_qb.p, tab, S_max, f16, stream, qkv_slabs.S, qkv_slabs.stride))) return rc;
                rope_done = true;
                rc = sm_llm_decode_attention_seg_tab(g->d_qb.p, tab, S, kv_max0 + j + 1, H, KV, dh, S_max, g->d_ctxb.p, f16, stream, c.llm_sliding_window);
                if (rc < 0) return rc;
                attn_done = rc == 0;                 // 1: contexts too long for the one-launch kernel -> the chunks below (RoPE + append are done)
            } else if (NC > 1) {           // SM_BIG_SEG streams per RoPE + append launch and per attention launch (pointer packs by value): 1 pack up to 128 streams, 4 at 512
                rope_done = attn_done = true;
                for (int b0 = 0; b0 < S; b0 += SM_BIG_SEG) {
                    const int bn = S - b0 < SM_BIG_SEG ? S - b0 : SM_BIG_SEG;
                    SmDecodeSegBig big;
                    for (int t = 0; t < bn; ++t) { big.kc[t] = act[b0 + t]->kc[l].p; big.vtc[t] = act[b0 + t]->vtc[l].p; big.pos[t] = act[b0 + t]->kv_len; }
                    char* qb = (char*)g->d_qb.p + (size_t)b0 * qn * 2;
                    if ((rc = sm_rope_kv_append_seg_big(g->d_qkvf.as<float>() + (size_t)b0 * (qn + 2 * kn), bn, H, KV, dh, m->rope_cos.as<float>(), m->rope_sin.as<float>(), qb, big, S_max, f16, stream))) return rc;
                    if (attn_done) {
                        rc = sm_llm_decode_attention_seg_big(qb, big, bn, H, KV, dh, S_max, (char*)g->d_ctxb.p + (size_t)b0 * qn * 2, f16, stream, c.llm_sliding_window);
                        if (rc < 0) return rc;
                        if (rc == 1) attn_done = false;          // contexts too long for the one-launch kernel -> every stream through the chunks below (RoPE + append are done)
                    }
                }
            }
            for (int ch = 0; ch < NC && !attn_done; ++ch) {
                SmDecodeSeg seg;
                for (int t = 0; t < cn(ch); ++t) { sm_stream* st = act[c0(ch) + t]; seg.kc[t] = st->kc[l].p; seg.vtc[t] = st->vtc[l].p; seg.pos[t] = st->kv_len; }
                char* qb = (char*)g->d_qb.p + (size_t)c0(ch) * qn * 2;
                if (!rope_done && (rc = sm_rope_kv_append_seg(g->d_qkvf.as<float>() + (size_t)c0(ch) * (qn + 2 * kn), cn(ch), H, KV, dh, m->rope_cos.as<float>(), m->rope_sin.as<float>(), qb, seg, S_max, f16, stream))) return rc;
                if ((rc = sm_llm_decode_attention_seg(qb, seg, cn(ch), H, KV, dh, S_max, g->d_ws.as<float>(), SM_DECODE_SPLITS, (char*)g->d_ctxb.p + (size_t)c0(ch) * qn * 2, f16, stream, c.llm_sliding_window))) return rc;
            }
            {   sm_linear_t a = lin(m, *w.o, g->d_ctxb.p, SM_X_BF16, S, qn);
                a.residual = x; a.ldr = ld; a.out_f32 = x; a.ldo = ld;
                if (!fuse_norm) { a.post_ln_gamma = w.ln2_w; a.post_ln_eps = c.llm_eps; a.post_ln_out = g->d_xnb.p; a.post_ln_ldo = ld; }
                if ((rc = sm_linear(&a, stream))) return rc; }
            if (S <= SM_MAX_SEG || (S <= 64 && (sm_skinny_lds64_on() >= 2 || (sm_skinny_lds64_on() == 1 && c.weights_fp8)))) {
                const Slot& gu = *w.gu;
                sm_linear_t a = fuse_norm ? lin(m, gu, x, SM_X_F32, S, ld) : lin(m, gu, g->d_xnb.p, SM_X_BF16, S, ld);
                if (fuse_norm) { a.norm_gamma = w.ln2_w; a.norm_eps = c.llm_eps; }
                a.N = c.llm_mlp;
                if (gu.fp8) { a.w2 = (const char*)gu.buf.p + (size_t)(c.llm_mlp / 16) * ((ld / 32 + 1) / 2) * 1024; a.w2_scale = gu.scale.as<float>() + c.llm_mlp; }
                else a.w2 = gu.buf.as<bf16_t>() + (size_t)(c.llm_mlp / 16) * (ld / 32) * 512;
                a.out_bf16 = g->d_actb.p; a.ldo_bf16 = c.llm_mlp;
                if ((rc = sm_linear(&a, stream))) return rc;
            } else {              // more rows than the dual weight-streaming kernel takes: gate | up as one tiled product with act_fn(gate) * up behind it
                                  // (in the 256 x 256 kernel's epilogue from ~440 streams, a SwiGLU pass inside sm_linear below that), as a prefill chunk does
                sm_linear_t a = lin(m, *w.gu, g->d_xnb.p, SM_X_BF16, S, ld);
                a.act = SM_ACT_SWIGLU_DUAL;
                a.out_bf16 = g->d_actb.p; a.ldo_bf16 = c.llm_mlp;
                if ((rc = sm_linear(&a, stream))) return rc;
            }
            {   sm_linear_t a = lin(m, *w.down, g->d_actb.p, SM_X_BF16, S, c.llm_mlp);
                a.residual = x; a.ldr = ld; a.out_f32 = x; a.ldo = ld;
                if (!fuse_norm) {
                    a.post_ln_gamma = l + 1 < c.llm_layers ? m->R.llm[l + 1].ln1_w : m->R.llm_norm;       // the next layer's input norm / the final norm
                    a.post_ln_eps = c.llm_eps; a.post_ln_ldo = ld;
                    if (fuse_rope && l + 1 < c.llm_layers) a.post_ln_out_f32 = g->d_xn32.as<float>();     // (the fused q/k/v + RoPE product reads fp32 rows)
                    else a.post_ln_out = g->d_xnb.p;
                }
                if ((rc = sm_linear(&a, stream))) return rc; }
        }
        for (sm_stream* s : act) s->kv_len += 1;
        if (!fuse_norm && c.llm_layers == 0 && (rc = sm_norm_ex(x, S, ld, ld, m->R.llm_norm, nullptr, c.llm_eps, 0, nullptr, g->d_xnb.p, ld, od, stream))) return rc;
        {   sm_linear_t a = fuse_norm ? lin(m, *m->R.lm_head, x, SM_X_F32, S, ld) : lin(m, *m->R.lm_head, g->d_xnb.p, SM_X_BF16, S, ld);
            if (fuse_norm) { a.norm_gamma = m->R.llm_norm; a.norm_eps = c.llm_eps; }
            a.out_f32 = g->d_log.as<float>(); a.ldo = V;
            if ((rc = sm_linear(&a, stream))) return rc; }
        if (NC > 1) {
            for (int b0 = 0; b0 < S; b0 += SM_BIG_SEG) {
                const int bn = S - b0 < SM_BIG_SEG ? S - b0 : SM_BIG_SEG;
                SmTokPtrsBig tok;
                for (int t = 0; t < bn; ++t) tok.p[t] = act[b0 + t]->next_tok.as<int32_t>();
                if ((rc = sm_argmax_rows_seg_big(g->d_log.as<float>() + (size_t)b0 * V, bn, V, V, tok, stream))) return rc;
            }
        } else
        for (int ch = 0; ch < NC; ++ch) {
            SmTokPtrs tok;
            for (int t = 0; t < cn(ch); ++t) tok.p[t] = act[c0(ch) + t]->next_tok.as<int32_t>();
            if ((rc = sm_argmax_rows_seg(g->d_log.as<float>() + (size_t)c0(ch) * V, cn(ch), V, V, tok, stream))) return rc;
        }
    }
    // each stream's own "last logits" as sm_llm_decode would have left them
    for (int t = 0; t < S; ++t)
        SM_HIP(hipMemcpyAsync(act[t]->lmlog.p, g->d_log.as<float>() + (size_t)t * V, (size_t)V * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SM_OK;
}
