"""Upper bound for threaded frame lanes: T host threads, each driving one-frame tower calls on its own HIP stream."""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig
os.environ["SM_VIT_SMALL_LANES"] = "1"
cfg = PathConfig(llm_layers=0, max_frames_per_call=8)
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
model.finalize()
frames = torch.randint(0, 256, (8, 336, 336, 3), dtype=torch.uint8, device="cuda")
N = 100
def run(T):
    streams = [torch.cuda.Stream() for _ in range(T)]
    def work(i, n):
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                model.vit_encode(frames[i:i + 1])
    for i in range(T): work(i, 3)
    torch.cuda.synchronize()
    th = [threading.Thread(target=work, args=(i, N)) for i in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"threads {T}: host issue {1e3*(t1-t0)/N:.3f} ms per round of {T} frames, complete {1e3*(t2-t0)/N:.3f} ms per round -> {T*N/(t2-t0):.0f} frames/s")
for T in (1, 2, 3, 4, 6, 8):
    run(T)
