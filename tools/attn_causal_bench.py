"""Micro-benchmark of the causal prefill attention (sm_llm_attention) at Mistral-7B's shape: n new tokens against a cache that holds pos0 older ones,
32 query heads over 8 K/V heads of 128.  HIP-event time per launch, causal FLOPs, max error against fp32 torch (SM_CHECK=1).
    SM_ATTN_PREFILL=1|0 [SM_ATTN_PAIR=1|0|2] python tools/attn_causal_bench.py [n] [pos0]
SM_ATTN_PREFILL=1 (default): the round-6 prefill kernel; 0: the general tile kernel, whose block schedule SM_ATTN_PAIR picks (1: paired tiles, 0: one tile per block in
tile order, 2: longest tiles first)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pos0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H, KV, dh = 32, 8, 128
S_max = (pos0 + n + 255) // 256 * 256
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(5)
q = (torch.randn(n, H * dh, device="cuda", generator=g) * 0.5).bfloat16()
kc = (torch.randn(S_max, KV * dh, device="cuda", generator=g) * 0.5).bfloat16()
v = (torch.randn(S_max, KV, dh, device="cuda", generator=g) * 0.5).bfloat16()
vtc = v.permute(1, 2, 0).contiguous()                      # [KV][dh][S_max]
ctx = torch.empty(n, H * dh, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream


def run():
    _lib.check(lib.sm_llm_attention(q.data_ptr(), kc.data_ptr(), vtc.data_ptr(), n, pos0, H, KV, dh, S_max, ctx.data_ptr(), st))


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
pairs = n * pos0 + n * (n + 1) / 2                          # visible (query, key) pairs
flops = 4.0 * H * dh * pairs
print(f"causal attention n={n} pos0={pos0} SM_ATTN_PREFILL={os.environ.get('SM_ATTN_PREFILL', '(1)')} SM_ATTN_PAIR={os.environ.get('SM_ATTN_PAIR', '(1)')}: {us:.1f} us  {flops / us / 1e6:.1f} TF/s")
if os.environ.get("SM_CHECK", "0") == "1":
    nk = pos0 + n
    qf = q.view(n, H, dh).transpose(0, 1).float()                                   # [H][n][dh]
    kf = kc[:nk].view(nk, KV, dh).transpose(0, 1).float().repeat_interleave(H // KV, 0)
    vf = v[:nk].transpose(0, 1).float().repeat_interleave(H // KV, 0)
    s = qf @ kf.transpose(-1, -2) / dh ** 0.5
    mask = torch.arange(nk, device="cuda")[None, :] > (pos0 + torch.arange(n, device="cuda"))[:, None]
    s.masked_fill_(mask[None], float("-inf"))
    ref = (torch.softmax(s, -1) @ vf).transpose(0, 1).reshape(n, H * dh)
    print("   max abs err vs fp32 torch:", (ctx.float() - ref).abs().max().item())
