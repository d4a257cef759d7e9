"""Connector + gate weight pass alone: N push_pooled calls of R rows, wall clock -> us per pass and the HBM fraction on 1.83 GB.
    python tools/pass_bench.py [R=28] [N=200]        FP8=1: weight-only fp8 gate weights (SM_FP8_LDS=0: the 17..32-row products expand them to bf16 per call, the round-4 form)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig
R = int(sys.argv[1]) if len(sys.argv) > 1 else 28
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
FP8 = int(os.environ.get("FP8", "0"))
cfg = PathConfig(llm_layers=0, max_frames_per_call=32, weights_fp8=FP8)
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
model.finalize()
s = model.open_stream(max_frames=N * R + 64 * R, max_seq=64)
pooled = torch.randn(R, cfg.vit_hidden, device="cuda")
for _ in range(10):
    s.push_pooled(pooled)
s.reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    s.push_pooled(pooled)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / N * 1e6
byts = bench.conn_gate_bytes(cfg) if hasattr(bench, "conn_gate_bytes") else 1.83e9
print({"rows": R, "weights_fp8": FP8, "us_per_pass": round(us, 1), "hbm_frac_on_1.83GB": round(1.83e9 / (us * 1e-6) / 8e12, 4)})
