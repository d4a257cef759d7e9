"""Self-synchronising / restart-interval GPU entropy decode against PIL on RANDOM files: sizes 8..720, qualities 3..100, 4:4:4 / 4:2:2 / 4:2:0 / grey, optimised
tables or not, restart intervals or not, four kinds of content.  Every byte of every frame is compared; then damaged variants of some files (cut short,
64 bytes overwritten) must be reported or decode to a frame of the right shape -- never hang, never touch memory outside the frame (the process survives).
    python tools/jpeg_fuzz.py [N=1500] [seed=1]"""
import os, sys, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import util_jpeg as U
from streammind_amd import native

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dec = native.JpegDecoder(threads=4)
dec.keep_sync_rounds = True


def content(w, h, kind, gray):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:
        img = rng.integers(0, 256, (h, w, 3))                                    # white noise: the longest codes
    elif kind == 1:
        img = np.stack([128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 128 + 110 * np.sin((xx + yy) / 13.0), 255.0 * ((xx // 11 + yy // 5) % 2)], axis=2) + rng.normal(0, 10, (h, w, 3))
    elif kind == 2:
        img = np.full((h, w, 3), rng.integers(0, 256, 3))                        # flat: DC + EOB only
    else:
        img = 255.0 * (((xx // int(rng.integers(2, 40))) + (yy // int(rng.integers(2, 40)))) % 2)[..., None] * np.ones(3) + rng.normal(0, 3, (h, w, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img[:, :, 0].copy() if gray else img


t0 = time.time()
bad, rounds_hist, n_gpu, damaged_reported, damaged_decoded, host_fallbacks, not_chained = [], {}, 0, 0, 0, 0, []
for i in range(N):
    w, h = int(rng.integers(8, 721)), int(rng.integers(8, 721))
    if rng.random() < 0.1:
        w, h = 1280, 720
    gray = rng.random() < 0.1
    kw = dict(quality=int(rng.integers(3, 101)))
    if not gray:
        kw["subsampling"] = int(rng.integers(0, 3))
    if rng.random() < 0.3:
        kw["optimize"] = True
    r = rng.random()
    if r < 0.15:
        kw["restart_marker_blocks"] = int(rng.integers(1, 40))
    elif r < 0.25:
        kw["restart_marker_rows"] = int(rng.integers(1, 4))
    img = content(w, h, int(rng.integers(0, 4)), gray)
    try:
        b = U.encode(img, **kw)
    except OSError:                                                               # (Pillow refuses some restart / optimize combinations)
        kw = {k: v for k, v in kw.items() if not k.startswith("restart") and k != "optimize"}
        b = U.encode(img, **kw)
    want = U.pil_decode(b)
    if os.environ.get("FUZZ_TRACE"):
        print("file", i, w, h, gray, kw, len(b), flush=True)
    before = dec.gpu_entropy_frames
    try:
        got = dec.decode([b], entropy="gpu")[0].cpu().numpy()
    except Exception as e:
        if "host path" in str(e):                                                 # a documented refusal (e.g. more than 4096 restart intervals): what "auto" does with it
            host_fallbacks += 1
            got = dec.decode([b], entropy="auto")[0].cpu().numpy()
        elif "did not settle" in str(e) and int(dec.last_entropy_status[0]) == 5:   # the records did not chain in 16 rounds (seen once: q100 noise, no end-of-block
            not_chained.append({"i": i, "w": w, "h": h, "kw": kw, "bytes": len(b)})  # anywhere, one block per subsequence): reported, "auto" decodes it on the host
            got = dec.decode([b], entropy="auto")[0].cpu().numpy()
        else:
            bad.append({"i": i, "w": w, "h": h, "kw": kw, "gray": gray, "error": str(e)[:200]})
            continue
    n_gpu += dec.gpu_entropy_frames - before
    if got.shape != want.shape or not np.array_equal(got, want):
        bad.append({"i": i, "w": w, "h": h, "kw": kw, "gray": gray, "bytes_differ": int((got != want).sum()) if got.shape == want.shape else -1})
    if "restart_marker_blocks" not in kw and "restart_marker_rows" not in kw and dec.last_sync_rounds:
        rounds_hist[dec.last_sync_rounds[0]] = rounds_hist.get(dec.last_sync_rounds[0], 0) + 1
    if i % 10 == 0 and len(b) > 600:                                              # damaged variants of every tenth file
        for variant in range(2):
            m = bytearray(b)
            if variant == 0:
                m = m[:len(m) * 2 // 3] + b"\xff\xd9"
            else:
                at = int(rng.integers(len(m) // 2, len(m) - 80))
                m[at:at + 64] = bytes(int(v) for v in rng.integers(0, 255, 64))
            if os.environ.get("FUZZ_TRACE"):
                print("  damaged variant", variant, flush=True)
            try:
                o = dec.decode([bytes(m)], entropy="gpu")
                assert tuple(o.shape) == (1,) + want.shape
                damaged_decoded += 1
            except AssertionError:
                raise
            except Exception:
                damaged_reported += 1
torch.cuda.synchronize()
print(json.dumps({"files": N, "seed": seed, "decoded_on_gpu": n_gpu, "refused_with_reason_then_host": host_fallbacks, "not_chained_in_16_rounds_then_host": not_chained[:4], "mismatches_or_errors": len(bad), "first_bad": bad[:6],
                  "rounds_until_the_records_chained": dict(sorted(rounds_hist.items())), "damaged_reported": damaged_reported, "damaged_decoded_to_a_frame": damaged_decoded,
                  "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
