"""LayerNorm folding (sm_linear_t.fold_*), shape by shape at the bench's lane size: the fp32 + residual products (out-proj, fc2) plain / as the
fold's PRODUCER, the 16-bit-output products (q|k|v, fc1) plain / as its CONSUMER, and the LayerNorm launch the fold removes -- HIP-event time per
launch (sm_prof_*), interleaved rounds on one box.
    gpurun -- 'python tools/fold_bench.py [frames=28]'"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import _lib, native  # noqa: E402

torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 28
M, D, F = B * 577, 1024, 4096
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s, std=1.0: torch.randn(*s, generator=g, device="cuda") * std
W = {"qkv": (3 * D, D), "out": (D, D), "fc1": (F, D), "fc2": (D, F)}
wp = {k: native.pack_weight(rn(n, kk, std=kk ** -0.5).bfloat16()) for k, (n, kk) in W.items()}
bias = {k: rn(n, std=0.1) for k, (n, kk) in W.items()}
x = rn(M, D)
xin = {D: rn(M, D).bfloat16(), F: rn(M, F).bfloat16()}
ht = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
stats = torch.zeros(M, D // 256, 2, device="cuda")
gam, bet = 1 + 0.1 * rn(D), 0.1 * rn(D)
gv = {k: rn(W[k][0]) for k in ("qkv", "fc1")}
out16 = {k: torch.empty(M, W[k][0], dtype=torch.bfloat16, device="cuda") for k in ("qkv", "fc1")}


def run(what):
    if what in ("out", "fc2"):
        k = W[what][1]
        native.linear(xin[k], wp[what], D, k, bias=bias[what], residual=x, out=x)
    elif what in ("out+fold", "fc2+fold"):
        w = what[:3].rstrip("+")
        k = W[w][1]
        native.linear(xin[k], wp[w], D, k, bias=bias[w], residual=x, out=x, post_ln=(gam, bet, 1e-5, ht), fold_out=stats)
    elif what in ("qkv", "fc1"):
        native.linear(ht, wp[what], W[what][0], D, bias=bias[what], act=1 if what == "fc1" else 0, out=out16[what])
    else:
        w = what[:3]
        native.linear(ht, wp[w], W[w][0], D, act=1 if w == "fc1" else 0, out=out16[w], fold_in=(stats, gv[w], bias[w], 1e-5))


def timed(what, n=20):
    for _ in range(3):
        run(what)
    torch.cuda.synchronize()
    lib.sm_prof_reset(); lib.sm_prof_enable(1)
    for _ in range(n):
        run(what)
    torch.cuda.synchronize()
    lib.sm_prof_enable(0)
    cnt, ms = C.c_int(), C.c_float()
    lib.sm_prof_read(0, C.byref(cnt), C.byref(ms))
    return ms.value / max(cnt.value, 1) * 1e3


native.linear(xin[D], wp["out"], D, D, bias=bias["out"], residual=x, out=x, post_ln=(gam, bet, 1e-5, ht), fold_out=stats)     # real statistics for the consumers
res = {}
for rnd in range(3):
    for what in ("out", "out+fold", "fc2", "fc2+fold", "qkv", "qkv+fold", "fc1", "fc1+fold"):
        res.setdefault(what, []).append(timed(what))
        x.normal_(generator=g)
print(f"frames {B} (M = {M}); us per launch, min of 3 interleaved rounds")
for what, v in res.items():
    print(f"  {what:9s} {min(v):7.1f}   all {[round(t, 1) for t in v]}")
