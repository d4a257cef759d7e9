"""Micro-benchmark of sm_vit_attention at the bench shape (B frames x 577 tokens, 16 heads x 64): HIP-event timing.
    python tools/attn_bench.py [B]"""
import os
import sys
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 28
S, H, dh = 577, 16, 64
lib = _lib.load()
qkv = (torch.randn(B * S, 3 * H * dh, device="cuda") * 0.5).bfloat16()
ctx = torch.empty(B * S, H * dh, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(lib.sm_vit_attention(qkv.data_ptr(), None, ctx.data_ptr(), B, S, H, dh, 0, 0, st))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
flops = 4.0 * B * H * S * S * dh
print(f"vit attention B={B}: {us:.1f} us  {flops / us / 1e6:.1f} TF/s")
if os.environ.get("SM_CHECK", "0") == "1":
    q, k, v = [t.view(B, S, H, dh).transpose(1, 2).float() for t in qkv.split(H * dh, dim=1)]
    ref = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v
    ref = ref.transpose(1, 2).reshape(B * S, H * dh)
    print("   max abs err vs fp32 torch:", (ctx.float() - ref).abs().max().item())
