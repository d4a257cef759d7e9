"""Race screen: the tiled GEMM (staggered wave groups, counted vmcnt), the LDS-DMA attention and the whole ViT are
deterministic, so repeated runs on the same inputs must be BIT-identical; any difference is a synchronisation bug.
    python tools/race_screen.py [seconds]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import native, _lib
import bench

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
lib = _lib.load()
torch.manual_seed(0)
cases = []
for (M, N, K, res, act, odt) in [(16156, 3072, 1024, False, 0, torch.bfloat16), (16156, 1024, 1024, True, 0, torch.float32),
                                  (16156, 4096, 1024, False, 1, torch.bfloat16), (16156, 1024, 4096, True, 0, torch.float32),
                                  (4616, 3072, 1024, False, 0, torch.bfloat16), (577, 1024, 4096, True, 0, torch.float32),
                                  (512, 4096, 14336, True, 0, torch.float32), (8078, 4096, 1024, False, 1, torch.bfloat16),
                                  (577, 3072, 1024, False, 0, torch.bfloat16), (577, 4096, 1024, False, 1, torch.bfloat16),
                                  (2308, 1024, 1024, True, 0, torch.float32), (328, 6144, 4096, False, 0, torch.float32)]:
    w = native.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16())
    x = torch.randn(M, K, device="cuda").bfloat16()
    r = torch.randn(M, N, device="cuda") if res else None
    b = torch.randn(N, device="cuda")
    cases.append((M, N, K, w, x, r, b, act, odt))
def run_gemm(c):
    M, N, K, w, x, r, b, act, odt = c
    return native.linear(x, w, N, K, bias=b, act=act, residual=r, out_dtype=odt)
refs = [run_gemm(c).clone() for c in cases]
B, S, H, dh = 28, 577, 16, 64
qkv = (torch.randn(B * S, 3 * H * dh, device="cuda") * 0.5).bfloat16()
ctx = torch.empty(B * S, H * dh, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
def run_attn():
    _lib.check(lib.sm_vit_attention(qkv.data_ptr(), None, ctx.data_ptr(), B, S, H, dh, 0, 0, st))
    return ctx
ref_attn = run_attn().clone()
from streammind_amd.native import NativeModel, PathConfig
cfg = PathConfig(llm_layers=0, max_frames_per_call=28)
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, seed=3)
model.finalize()
frames = bench.synthetic_frames_gpu(28, 336, 5, 0)
ref_vit = model.vit_encode(frames).clone()
# round 6: the folded tower (fp16 operands: LayerNorm statistics written by one product's epilogue, prefetched a tile ahead by the next product's persistent loop) and the
# causal prefill kernel (K / V^T tiles by LDS-DMA, one barrier per key tile, fragment batches pinned by sched_barriers)
cfg16 = PathConfig(llm_layers=0, max_frames_per_call=28, vit_fp16=True)
model16 = NativeModel(cfg16)
bench.random_weights_into(model16, cfg16, seed=3)
model16.finalize()
ref_vit16 = model16.vit_encode(frames).clone()
PN, PH, PKV, PD = 2048, 32, 8, 128
pq = (torch.randn(PN, PH * PD, device="cuda") * 0.5).bfloat16()
pk = (torch.randn(PN, PKV * PD, device="cuda") * 0.5).bfloat16()
pvt = (torch.randn(PKV, PD, PN, device="cuda") * 0.5).bfloat16()
pctx = torch.empty(PN, PH * PD, device="cuda", dtype=torch.bfloat16)
def run_prefill_attn():
    _lib.check(lib.sm_llm_attention(pq.data_ptr(), pk.data_ptr(), pvt.data_ptr(), PN, 0, PH, PKV, PD, PN, pctx.data_ptr(), st))
    return pctx
ref_pattn = run_prefill_attn().clone()
t0, it, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    for c, ref in zip(cases, refs):
        if not torch.equal(run_gemm(c), ref):
            bad += 1; print("GEMM mismatch", c[:3], "iteration", it, flush=True)
    if not torch.equal(run_attn(), ref_attn):
        bad += 1; print("attention mismatch, iteration", it, flush=True)
    if it % 4 == 0 and not torch.equal(model.vit_encode(frames), ref_vit):
        bad += 1; print("ViT mismatch, iteration", it, flush=True)
    if it % 4 == 2 and not torch.equal(model16.vit_encode(frames), ref_vit16):
        bad += 1; print("folded fp16 ViT mismatch, iteration", it, flush=True)
    if not torch.equal(run_prefill_attn(), ref_pattn):
        bad += 1; print("prefill attention mismatch, iteration", it, flush=True)
    it += 1
torch.cuda.synchronize()
print(f"race screen: {it} iterations x ({len(cases)} GEMMs + ViT attention + causal prefill attention) + {it // 4 + 1} ViT batches (bf16) + {(it + 1) // 4} (fp16, LayerNorms folded), {bad} mismatches")
sys.exit(1 if bad else 0)
