"""Per-block phase stamps of the 128x128 GEMM (variant library built with -DSM_GEMM128_TIMELINE):
    tools/build_variant.sh tl -DSM_GEMM128_TIMELINE && STREAMMIND_HIP_LIB=streammind_amd/lib/libstreammind_hip_tl.so python tools/gemm128_timeline.py [M]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streammind_amd import native, _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 577
lib = _lib.load()
raw = ctypes.CDLL(os.environ["STREAMMIND_HIP_LIB"])
tl = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
raw.sm_debug_set_timeline128.argtypes = [ctypes.c_void_p]
assert raw.sm_debug_set_timeline128(tl.data_ptr()) == 0
for name, N, K, dt in (("qkv", 3072, 1024, torch.bfloat16), ("out", 1024, 1024, torch.float32), ("fc1", 4096, 1024, torch.bfloat16), ("fc2", 1024, 4096, torch.float32)):
    w = native.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16())
    x = torch.randn(M, K, device="cuda").bfloat16()
    junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    for it in range(3):
        if os.environ.get("COLD", "1") == "1":
            junk.fill_(it)                    # evict the weights: a tick touches each layer's weights once
            if os.environ.get("WARMX", "0") == "1":
                x.add_(0)                     # ... but x was just written by the previous kernel
        tl.zero_(); torch.cuda.synchronize()
        native.linear(x, w, N, K, out_dtype=dt)
        torch.cuda.synchronize()
    h = tl.cpu().view(-1, 8)
    h = h[h[:, 0] > 0]
    t0 = int(h[:, 0].min())
    us = (h[:, :4] - t0).double() * 0.01
    q = lambda c, p: float(us[:, c].quantile(p))
    print(f"{name} M={M} N={N} K={K}: {h.shape[0]} blocks; span {float(us[:, 3].max()):.1f} us; start p0/p50/p100 {q(0,0):.1f}/{q(0,.5):.1f}/{q(0,1):.1f}; "
          f"first tile landed +{float((us[:,1]-us[:,0]).mean()):.2f}; loop {float((us[:,2]-us[:,1]).mean()):.2f} (max {float((us[:,2]-us[:,1]).max()):.2f}); "
          f"epilogue {float((us[:,3]-us[:,2]).mean()):.2f} (max {float((us[:,3]-us[:,2]).max()):.2f}); end p50/p100 {q(3,.5):.1f}/{q(3,1):.1f}")
