"""Micro-benchmark of the weight-streaming linear on the event-gate shapes (fp32 activations, hi/lo split).
    python tools/skinny_bench.py [M] [precise 0|1] [x bf16|f32]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import native

M = int(sys.argv[1]) if len(sys.argv) > 1 else 28
precise = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
xdt = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float32
shapes = [("v", 1024, 4096, False), ("o", 4096, 4096, False), ("qkv", 6144, 4096, False), ("gate_up", 14336, 4096, True), ("down", 4096, 14336, False),
          ("lm_head", 32000, 4096, False)]
for name, N, K, dual in shapes:
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    wp = native.pack_weight(w)
    w2p = native.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()) if dual else None
    x = torch.randn(M, K, device="cuda").to(xdt)
    # rotate over several weight copies so that the stream comes from HBM, not from the 256 MB MALL
    copies = [wp] + [wp.clone() for _ in range(max(0, min(7, (1 << 30) // (N * K * 2))))]
    for _ in range(3):
        native.linear(x, wp, N, K, w2p=w2p, precise=precise)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    e0.record()
    for i in range(n):
        native.linear(x, copies[i % len(copies)], N, K, w2p=w2p, precise=precise)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    mb = N * K * 2 * (2 if dual else 1) / 1e6
    print(f"{name:8s} M={M} N={N} K={K} dual={dual}: {us:7.1f} us  {mb / us:.2f} TB/s of weights (launch pair where the K slices are summed by a second kernel)", flush=True)
