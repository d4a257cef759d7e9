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
t0, it, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    for c, ref in zip(cases, refs):
        if not torch.equal(run_gemm(c), ref):
            bad += 1; print("GEMM mismatch", c[:3], "iteration", it, flush=True)
    if not torch.equal(run_attn(), ref_attn):
        bad += 1; print("attention mismatch, iteration", it, flush=True)
    if it % 4 == 0 and not torch.equal(model.vit_encode(frames), ref_vit):
        bad += 1; print("ViT mismatch, iteration", it, flush=True)
    it += 1
torch.cuda.synchronize()
print(f"race screen: {it} iterations x ({len(cases)} GEMMs + attention) + {it // 4 + 1} ViT batches, {bad} mismatches")
sys.exit(1 if bad else 0)
