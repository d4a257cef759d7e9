"""Is the weight-streaming GEMV faster when its weights sit in the 256 MB Infinity Cache?  Run under
`rocprofv3 --kernel-trace --stats`: mode `hot` re-uses one weight copy, `cold` rotates over enough copies to exceed the cache.
    python tools/mall_probe.py hot|cold [M]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import native

mode = sys.argv[1]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for name, N, K in [("o", 4096, 4096), ("qkv", 6144, 4096), ("down", 4096, 14336)]:
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    wp = native.pack_weight(w)
    ncopy = 1 if mode == "hot" else (1 << 30) // (N * K * 2)
    copies = [wp] + [wp.clone() for _ in range(ncopy - 1)]
    x = torch.randn(M, K, device="cuda").bfloat16()
    for i in range(64):
        native.linear(x, copies[i % len(copies)], N, K, precise=False)
    torch.cuda.synchronize()
