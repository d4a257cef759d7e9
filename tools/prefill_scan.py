"""Prefill time against the number of new tokens (Mistral-7B widths, empty cache): where the row-count dispatch has steps.   python tools/prefill_scan.py [n,n,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from streammind_amd.native import NativeModel, PathConfig
sizes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 4, 8, 9, 12, 13, 16, 17, 24, 32, 33, 48, 64, 65, 96, 128, 129, 160, 191, 192, 256, 257, 328, 512, 513, 768, 1024]
cfg = PathConfig(llm_layers=32, max_frames_per_call=1, vit_layers=2, weights_fp8=int(os.environ.get("FP8", "0")))
m = NativeModel(cfg); bench.random_weights_into(m, cfg, 1); bench.random_llm_weights_into(m, cfg, 2); m.finalize()
s = m.open_stream(max_frames=8, max_seq=2048)
g = torch.Generator(device="cuda").manual_seed(3)
prev = None
for n in sizes:
    ids = torch.randint(3, cfg.llm_vocab, (n,), generator=g, device="cuda", dtype=torch.int32)
    for _ in range(2):
        s.set_kv_len(0); s.prefill(ids)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        s.set_kv_len(0); s.prefill(ids)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(n, round(ms, 3), "ms", round(n / ms, 1), "tokens/ms", "" if prev is None else "per extra token %.4f ms" % ((ms - prev[1]) / (n - prev[0])), flush=True)
    prev = (n, ms)
