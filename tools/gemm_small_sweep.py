"""Sweep of the 128x128 tiled GEMM on small-M shapes: ring (SM_GEMM_RING) x 8-wave blocks (SM_GEMM_W8) x split-K (SM_SPLITK) per shape.
    python tools/gemm_small_sweep.py            # spawns one process per (ring, S) and prints a table
    python tools/gemm_small_sweep.py worker     # one configuration (reads the env), prints "name us" lines"""
import os, sys, subprocess, collections
SHAPES = [("vit_qkv", 577, 3072, 1024), ("vit_out", 577, 1024, 1024), ("vit_fc1", 577, 4096, 1024), ("vit_fc2", 577, 1024, 4096),
          ("vit2_qkv", 1154, 3072, 1024), ("vit2_out", 1154, 1024, 1024), ("vit2_fc1", 1154, 4096, 1024), ("vit2_fc2", 1154, 1024, 4096),
          ("vit4_qkv", 2308, 3072, 1024), ("vit4_out", 2308, 1024, 1024), ("vit4_fc1", 2308, 4096, 1024), ("vit4_fc2", 2308, 1024, 4096),
          ("llm328_qkv", 328, 6144, 4096), ("llm328_o", 328, 4096, 4096), ("llm328_gu", 328, 28672, 4096), ("llm328_down", 328, 4096, 14336),
          ("llm1k_qkv", 1024, 6144, 4096), ("llm1k_o", 1024, 4096, 4096), ("llm1k_down", 1024, 4096, 14336)]
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from streammind_amd import native
    for name, M, N, K in SHAPES:
        w = native.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16())
        x = torch.randn(M, K, device="cuda").bfloat16()
        res = torch.randn(M, N, device="cuda")
        for _ in range(3):
            native.linear(x, w, N, K, out_dtype=torch.float32)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            native.linear(x, w, N, K, out_dtype=torch.float32)
        e1.record(); torch.cuda.synchronize()
        print(name, e0.elapsed_time(e1) * 50, flush=True)
    sys.exit(0)
tab = collections.defaultdict(dict)
cfgs = [(r, w, s) for r in (0, 1) for w in (0, 2) for s in (1, 2, 4)]
for r, w, s in cfgs:
    env = dict(os.environ, SM_GEMM_RING=str(r), SM_GEMM_W8=str(w), SM_SPLITK=str(s))
    out = subprocess.run([sys.executable, __file__, "worker"], env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        n, us = line.split()
        tab[n][(r, w, s)] = float(us)
print("shape".ljust(12) + "".join(f" r{r}w{w}s{s}".rjust(8) for r, w, s in cfgs))
for name, M, N, K in SHAPES:
    row = tab[name]
    best = min(row, key=row.get) if row else None
    print(name.ljust(12) + "".join(f"{row.get(c, float('nan')):8.1f}" for c in cfgs) + f"  best r{best[0]}w{best[1]}s{best[2]} tiles {((M + 127) // 128) * ((N + 127) // 128)}")
