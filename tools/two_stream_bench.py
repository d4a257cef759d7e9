"""Experiment: independent video streams of ONE sm_model (NMODELS=1, default; the tower's workspaces are per HIP stream) or of
separate replicas (NMODELS=0: one model per stream, the round-1 form) driven on different HIP streams of ONE GPU, to see
whether the hardware overlaps one stream's HBM-bound phases (epilogues, LayerNorm, gate GEMVs) with the other's
MFMA-bound GEMM main loops.   python tools/two_stream_bench.py [batch] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 28
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NS = int(os.environ.get("NSTREAMS", "2"))
cfg = PathConfig(llm_layers=0, max_frames_per_call=B)
models, streams, hs = [], [], []
ONE = os.environ.get("NMODELS", "1") == "1"
PIPE = os.environ.get("PIPE", "0") == "1"          # sm_stream_push_frames_pipelined (gate pass of call i under the tower of call i+1)
for i in range(NS):
    if i == 0 or not ONE:
        m = NativeModel(cfg); bench.random_weights_into(m, cfg, 1 + i); m.finalize()
    models.append(m); streams.append(m.open_stream(max_frames=B * (steps + 4) + 16, max_seq=64)); hs.append(torch.cuda.Stream())
frames = bench.synthetic_frames_gpu(B * 4, 336, 1, 0)
torch.cuda.synchronize()
def step(i):
    for k in range(NS):
        with torch.cuda.stream(hs[k]):
            (streams[k].push_frames_pipelined if PIPE else streams[k].push_frames)(frames[(i % 4) * B:(i % 4) * B + B])
for i in range(2): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps): step(i)
if PIPE:
    for st in streams: st.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"NSTREAMS={NS} B={B}: {NS * B * steps / dt:.1f} frames/s aggregate, {dt / steps * 1e3:.2f} ms per round")
