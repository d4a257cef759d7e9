"""Self-synchronising JPEG decode: rounds until the states settle and batch time, over qualities / chroma layouts (diagnostic)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import util_jpeg as U
from streammind_amd import native
dec = native.JpegDecoder(threads=16)
dec.keep_sync_rounds = True
out = []
for (w, h, q, ss, n) in ((1280, 720, 85, 2, 28), (1280, 720, 25, 2, 28), (1280, 720, 98, 0, 8), (1280, 720, 98, 2, 28), (1920, 1080, 75, 1, 8), (1280, 720, 60, 2, 112)):
    fr = [U.encode(U.test_image(w, h, 40 + i), quality=q, subsampling=ss) for i in range(min(n, 28))] * max(1, n // 28)
    try:
        o = dec.decode(fr, entropy="gpu")
        ok = bool(np.array_equal(o[0].cpu().numpy(), U.pil_decode(fr[0])))
        st = "ok"
    except Exception as e:
        ok, st = False, str(dec.last_entropy_status.tolist())
    rounds = dec.last_sync_rounds
    dec.keep_sync_rounds = False
    t = None
    if st == "ok":
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            dec.decode(fr, entropy="gpu")
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    dec.keep_sync_rounds = True
    out.append({"case": f"{len(fr)} x {w}x{h} q{q} ss{ss}", "kb_per_frame": round(sum(map(len, fr)) / len(fr) / 1e3, 1), "pil_equal": ok, "status": st,
                "rounds_max": max(rounds) if rounds else None, "rounds_mean": round(float(np.mean(rounds)), 1) if rounds else None,
                "ms_per_batch": round(t * 1e3, 2) if t else None, "frames_per_s": round(len(fr) / t, 1) if t else None})
    print(json.dumps(out[-1]), flush=True)
