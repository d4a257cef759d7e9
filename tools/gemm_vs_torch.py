"""Context number: torch.matmul (hipBLASLt / rocBLAS behind PyTorch-ROCm) on the four ViT GEMM shapes next to sm_linear
(same inputs, bf16 output, no bias / residual), HIP-event timed.   python tools/gemm_vs_torch.py [M]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import native
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16156
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K in [("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096), ("sq4k", 4096, 4096)]:
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    wp = native.pack_weight(w)
    t_sm = timeit(lambda: native.linear(x, wp, N, K, out_dtype=torch.bfloat16))
    wt = w.t().contiguous()
    t_t1 = timeit(lambda: torch.matmul(x, wt))
    t_t2 = timeit(lambda: torch.nn.functional.linear(x, w))
    fl = 2 * M * N * K / 1e6
    print(f"{name:5s} M={M} N={N} K={K}: sm_linear {t_sm:7.1f} us {fl / t_sm:7.1f} TF/s | torch.matmul {t_t1:7.1f} us {fl / t_t1:7.1f} TF/s | F.linear {t_t2:7.1f} us {fl / t_t2:7.1f} TF/s", flush=True)
