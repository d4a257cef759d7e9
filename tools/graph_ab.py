"""hipGraph A/B (VERDICT r1 #5 / BASELINE configs[4] "hipGraph-captured per-frame gate step"): the SAME launch sequences issued
eagerly and replayed from a captured graph, on one box, interleaved.

  (a) one Mistral-7B decode step (sm_llm_decode(1): ~290 launches) at a fixed cache position
  (b) the per-frame gate step: one frame through ViT + connector + gate (sm_stream_push_frames(1): ~250 launches)
  (c) the same for 28 frames (the bench's step)

Capture goes through torch.cuda.CUDAGraph on the stream the library launches on (the library never synchronises or allocates
on a hot call after warm-up).  The host-side position counters (token row, kv_len) are baked into a captured graph, so each
replay rewrites the same token row / KV slot: identical work, which is what an A/B needs; a production graph path would need
device-side counters -- worth building only if this shows a win.  Writes profiles-ready JSON to stdout.
    gpurun -- 'python tools/graph_ab.py > gpurun_out/graph_ab.json'"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                    # noqa: E402
from streammind_amd.native import NativeModel, PathConfig       # noqa: E402

torch.set_grad_enabled(False)
FP8 = os.environ.get("FP8", "0") == "1"        # BASELINE configs[4]: gate + LLM weights in fp8 (weights_fp8 = 2), 16 frames per gate pass
cfg = PathConfig(llm_layers=32, max_frames_per_call=28, weights_fp8=2 if FP8 else 0)
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
bench.random_llm_weights_into(model, cfg, 2)
model.finalize()
lib = model.lib
frames = bench.synthetic_frames_gpu(28, 336, 1, 0)


def timeit(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def ab(name, make_eager, reps, rounds=5):
    """make_eager() -> fn issuing the launch sequence once at FIXED host state"""
    eager = make_eager()
    side = torch.cuda.Stream()
    for _ in range(3):
        eager()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        eager()                                                  # warm-up on the capture stream (lazy per-stream workspaces)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            eager()
    torch.cuda.synchronize()
    te, tg = [], []
    for _ in range(rounds):
        te.append(timeit(eager, reps))
        with torch.cuda.stream(side):
            tg.append(timeit(g.replay, reps))
    return {"what": name, "eager_us": round(min(te), 1), "graph_us": round(min(tg), 1), "eager_us_all": [round(v, 1) for v in te],
            "graph_us_all": [round(v, 1) for v in tg], "graph_over_eager": round(min(tg) / min(te), 4)}


out = []
s = model.open_stream(max_frames=256, max_seq=1024)
s.write_tokens(0, torch.randn(64, cfg.conn_d_model, device="cuda"))
ids = torch.cat([torch.randint(3, 32000, (64,), dtype=torch.int32, device="cuda"), -(torch.arange(64, dtype=torch.int32, device="cuda") + 1),
                 torch.randint(3, 32000, (200,), dtype=torch.int32, device="cuda")]).contiguous()
s.prefill(ids)
s.decode(8)
kv0 = s.kv_len
buf = torch.empty(1, dtype=torch.int32, device="cuda")


def mk_decode():
    def f():
        s.set_kv_len(kv0)                                        # host counter only: every issue appends at the same position
        lib.sm_llm_decode(s.h, 1, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return f


out.append(ab(f"Mistral-7B decode step at context {kv0} (sm_llm_decode, 1 token)", mk_decode, 40))

for B in (1, 28):
    st = model.open_stream(max_frames=64, max_seq=64)
    lg = torch.empty(B, 2, device="cuda")
    dc = torch.empty(B, dtype=torch.int32, device="cuda")
    fr = frames[:B].contiguous()

    def mk_push(st=st, lg=lg, dc=dc, fr=fr, B=B):
        def f():
            st.reset() if False else None
            lib.sm_stream_reset(st.h, torch.cuda.current_stream().cuda_stream)      # T = 0: every issue writes token rows 0..B-1
            lib.sm_stream_push_frames(st.h, fr.data_ptr(), B, lg.data_ptr(), dc.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return f
    out.append(ab(f"per-frame gate step: {B} frame(s) through ViT + connector + gate (sm_stream_push_frames)", mk_push, 30 if B == 1 else 10))

print(json.dumps({"tool": "tools/graph_ab.py", "weights": "fp8 e4m3 gate + LLM (weights_fp8 = 2)" if FP8 else "bf16", "note": "min over 5 interleaved rounds, microseconds per issue; graph = torch.cuda.CUDAGraph replay of the "
                  "identical launch sequence on a side stream", "results": out}, indent=1))
