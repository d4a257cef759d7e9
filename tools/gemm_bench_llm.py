"""tiled-GEMM micro-benchmark on the Mistral-7B prefill / teacher-forced shapes.   python tools/gemm_bench_llm.py [M]
(SM_GEMM_TILE=128|256128|256 forces a tile variant; FP8=1: fp8 weight image + fp8 x fp8 MFMA, FP8=wo: weight-only fp8; COLD=1: every launch multiplies ANOTHER copy of
the weights, > 600 MB in rotation, so that they come from HBM as in a decode step or a prefill -- the default re-uses one copy, which then lives in the 256 MB Infinity Cache)"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import native, _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = _lib.load()
tot = 0.0
for name, N, K in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336), ("lm_head", 32000, 4096)]:
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    F8 = os.environ.get("FP8", "")
    kw = {}
    if F8:
        wp, sc = native.pack_weight_fp8(w)
        kw = dict(w_scale=sc, fp8_mfma=(F8 == "1"))
    else:
        wp = native.pack_weight(w)
    del w
    copies = [wp]
    if os.environ.get("COLD") == "1" and not F8:
        copies += [wp.clone() for _ in range(max(1, int(600e6 // (N * K * 2))))]
    for i in range(2):
        native.linear(x, copies[i % len(copies)], N, K, out_dtype=torch.float32, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10):
        native.linear(x, copies[i % len(copies)], N, K, out_dtype=torch.float32, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    del copies
    tot += us * (1 if name == "lm_head" else 32)
    print(f"{name:8s} M={M} N={N} K={K}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF/s  weights at {N * K * 2 / us / 1e6:5.2f} TB/s", flush=True)
print(f"32 layers + head: {tot / 1e3:.2f} ms")
