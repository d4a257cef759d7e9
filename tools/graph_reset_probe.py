"""Debug probe: captured sm_stream_reset (+ push) at full size after other captures happened in the process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig
torch.set_grad_enabled(False)

def capture(fn):
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            fn()
    torch.cuda.synchronize()
    return g, side

keep = []
for fp8 in (0, 2):
    cfg = PathConfig(llm_layers=0 if not fp8 else 32, max_frames_per_call=16, weights_fp8=fp8)
    m = NativeModel(cfg)
    bench.random_weights_into(m, cfg, 1)
    if fp8:
        bench.random_llm_weights_into(m, cfg, 2)
    m.finalize()
    lib = m.lib
    for B in (1, 16, 1):
        frames = bench.synthetic_frames_gpu(3 * B, 336, 1, 0)
        slot = frames[:B].clone()
        st = m.open_stream(max_frames=32, max_seq=512 if fp8 else 64)
        lg = torch.empty(B, 2, device="cuda"); dc = torch.empty(B, dtype=torch.int32, device="cuda")
        def reset_only():
            assert lib.sm_stream_reset(st.h, torch.cuda.current_stream().cuda_stream) == 0
        def step():
            cs = torch.cuda.current_stream().cuda_stream
            assert lib.sm_stream_reset(st.h, cs) == 0
            assert lib.sm_stream_push_frames(st.h, slot.data_ptr(), B, lg.data_ptr(), dc.data_ptr(), cs) == 0
        g0, side0 = capture(reset_only)
        bad0 = 0
        for r in range(5):
            for t in st.state():
                t.fill_(1.0)
            torch.cuda.synchronize()
            with torch.cuda.stream(side0):
                g0.replay()
            torch.cuda.synchronize()
            bad0 += sum(int((t != 0).sum()) for t in st.state())
        g, side = capture(step)
        ref = m.open_stream(max_frames=32, max_seq=512 if fp8 else 64)
        res = []
        for r in range(6):
            slot.copy_(frames[(r % 3) * B:(r % 3 + 1) * B]); torch.cuda.synchronize()
            with torch.cuda.stream(side):
                g.replay()
            torch.cuda.synchronize()
            ref.reset()
            wl, wd = ref.push_frames(frames[(r % 3) * B:(r % 3 + 1) * B].contiguous())
            torch.cuda.synchronize()
            ds = [int((a != b).sum()) for a, b in zip(st.state(), ref.state())]
            res.append((bool(torch.equal(lg, wl)), ds))
        print(f"fp8={fp8} B={B}: reset-only replay left {bad0} non-zero state words; replays (logits equal, state diffs): {res}", flush=True)
        keep.append((g0, g, st, ref))
