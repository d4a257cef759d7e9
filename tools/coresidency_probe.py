"""Does a 16-register one-wave kernel run UNDERNEATH a 256 x 256 GEMM of another HIP stream (in the 16 VGPRs per SIMD its two
248-register waves leave free), or do the two time-slice?  Stream A: n back-to-back ViT GEMMs (M = 16156); stream B: n LayerNorm
passes over [16156][1024] fp32 -- as the light kernel (SM_NORM_LIGHT=1, ln_light_kernel) or as the ordinary one (51 VGPRs).
Wall clock of A alone, B alone, and both at once: co-residency shows as  both ~ max(A, B),  time slicing as  both ~ A + B.
    SM_NORM_LIGHT=1 python tools/coresidency_probe.py ; SM_NORM_LIGHT=0 python tools/coresidency_probe.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import _lib, native       # noqa: E402

lib = _lib.load()
M, D = 16156, 1024
n = 60
x32 = torch.randn(M, D, device="cuda")
xn = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
g, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def norms(k):
    with torch.cuda.stream(sb):
        for _ in range(k):
            _lib.check(lib.sm_norm_ex(x32.data_ptr(), M, D, D, g.data_ptr(), b.data_ptr(), C.c_float(1e-5), 0, None, xn.data_ptr(), D, 0, sb.cuda_stream))


def wall(fa, fb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa:
        fa()
    if fb:
        fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for name, N, K, act in [("qkv (persistent, 244 VGPRs, 160 KiB LDS)", 3072, 1024, 0), ("fc1 (persistent + quick_gelu, 238 VGPRs -> 240 allocated)", 4096, 1024, 1),
                        ("fc2 (fp32 + residual, 242 VGPRs, 128 KiB LDS)", 1024, 4096, 0)]:
    w = native.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16())
    xa = torch.randn(M, K, device="cuda").bfloat16()
    res = torch.zeros(M, N, device="cuda") if N == 1024 else None

    def gemms(k, res=res, xa=xa, w=w, N=N, K=K, act=act):
        with torch.cuda.stream(sa):
            for _ in range(k):
                if res is not None:
                    native.linear(xa, w, N, K, residual=res, out=res)
                else:
                    native.linear(xa, w, N, K, act=act, out_dtype=torch.bfloat16)

    gemms(3); norms(3)
    ta = min(wall(lambda: gemms(n), None) for _ in range(3))
    # as many norm launches as fill the same wall clock alone, so both streams stay busy for the whole concurrent run
    tb1 = min(wall(None, lambda: norms(n)) for _ in range(3))
    kb = max(1, int(round(n * ta / tb1)))
    tb = min(wall(None, lambda: norms(kb)) for _ in range(3))
    tab = min(wall(lambda: gemms(n), lambda: norms(kb)) for _ in range(3))
    print(f"{name}: GEMM alone {ta / n:7.1f} us/launch ({n}), norm alone {tb / kb:7.1f} us/launch ({kb}), both {tab:9.0f} us vs A {ta:9.0f} + B {tb:9.0f} = {ta + tb:9.0f}"
          f" -> overlap {100 * (ta + tb - tab) / min(ta, tb):5.1f} % of the shorter one (SM_NORM_LIGHT={os.environ.get('SM_NORM_LIGHT', '0')})", flush=True)
