"""Mistral-7B decode-only micro-benchmark (for rocprofv3 kernel traces of the decode step)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig

cfg = PathConfig(llm_layers=32, max_frames_per_call=1, vit_layers=2, weights_fp8=os.environ.get("FP8", "0") == "1", llm_fp16=os.environ.get("FP16", "0") == "1")    # short ViT: perception is not measured here; FP8=1: weight-only fp8 mode; FP16=1: fp16 operands (llm_fp16)
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
bench.random_llm_weights_into(model, cfg, 2)
model.finalize()
s = model.open_stream(max_frames=16384, max_seq=int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
nctx = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256]        # frame tokens of the context (+ 72 text tokens)
for n in nctx:
    r = bench.decode_leg(model, s, cfg, n_ctx_frames=n, n_new=int(sys.argv[1]) if len(sys.argv) > 1 else 64)
    print(r if len(nctx) == 1 else {k: r[k] for k in ("context_tokens", "tokens_per_s", "ms_per_token")})
