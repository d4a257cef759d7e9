"""Micro-benchmark of sm_linear (tiled GEMM) on ViT/LLM shapes: HIP-event timing of back-to-back launches.
    python tools/gemm_bench.py [M]            (env SM_GEMM_VARIANT=1|2 selects the ablation builds)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import native

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4616
shapes = [("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096), ("sq4k", 4096, 4096)]
CHECK = os.environ.get("SM_CHECK", "0") == "1"
for name, N, K in shapes:
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    wp = native.pack_weight(w)
    for _ in range(3):
        y = native.linear(x, wp, N, K, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    if CHECK:
        ref = (x[:512].float() @ w.float().t())
        err = ((y[:512].float() - ref).abs().max() / ref.abs().max()).item()
        ref2 = (x[-300:].float() @ w.float().t())
        err2 = ((y[-300:].float() - ref2).abs().max() / ref2.abs().max()).item()
        print(f"   check rel err {err:.2e} {err2:.2e}")
    import ctypes as C
    from streammind_amd import _lib
    lib = _lib.load()
    n = 30
    lib.sm_prof_reset(); lib.sm_prof_enable(1)
    for _ in range(n):
        y = native.linear(x, wp, N, K, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    lib.sm_prof_enable(0)
    cnt, ms = C.c_int(), C.c_float()
    lib.sm_prof_read(0, C.byref(cnt), C.byref(ms))
    us = ms.value / cnt.value * 1e3
    print(f"{name:5s} M={M} N={N} K={K}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
