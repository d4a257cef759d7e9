"""Launch floor and per-k-tile slope of the 128x128 tiled GEMM at one ViT frame (M = 577): time vs K.  python tools/gemm_kslope.py [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import native
M, N = 577, int(sys.argv[1]) if len(sys.argv) > 1 else 3072
for K in (64, 128, 256, 512, 1024, 2048):
    w = native.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16())
    x = torch.randn(M, K, device="cuda").bfloat16()
    for _ in range(3):
        native.linear(x, w, N, K, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        native.linear(x, w, N, K, out_dtype=torch.bfloat16)
    e1.record(); torch.cuda.synchronize()
    print(f"N={N} K={K:5d} k-tiles={K // 64:3d}: {e0.elapsed_time(e1) * 20:7.2f} us", flush=True)
