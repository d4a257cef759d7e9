"""Mint tests/golden/g19_jpeg.npz: small baseline JPEGs (every sampling mode the decoder implements, an odd size, a grey image, restart
intervals, a frame without Huffman tables) together with the RGB frames PIL / libjpeg-turbo decodes them to.  The fixture pins the
JPEG path (C-ABI host entropy decoder + oracle/jpeg_oracle.py + the HIP kernels) against the decoder the reference's frame loading
uses (decord / PIL -> libjpeg), independently of the PIL build present when the tests run.
    python oracle/make_jpeg_golden.py          (PIL %s with libjpeg-turbo; data only: JPEG byte strings and decoded arrays)"""
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util_jpeg as U          # noqa: E402  (image generator + encoder settings shared with the live tests)

CASES = [(64, 48, False, dict(quality=75, subsampling=2)), (47, 33, False, dict(quality=85, subsampling=1)), (40, 24, False, dict(quality=90, subsampling=0)),
         (33, 21, True, dict(quality=80)), (96, 64, False, dict(quality=60, subsampling=2, restart_marker_blocks=5))]


def main():
    import PIL
    from PIL import features
    out = {"pil_version": np.array(PIL.__version__), "libjpeg": np.array(str(features.version("jpg")) + (" turbo" if features.check_feature("libjpeg_turbo") else ""))}
    for i, (w, h, gray, kw) in enumerate(CASES):
        b = U.encode(U.test_image(w, h, 300 + i, gray), **kw)
        out[f"jpeg{i}"] = np.frombuffer(b, dtype=np.uint8)
        out[f"rgb{i}"] = U.pil_decode(b)
    bare = U.strip_dht(bytes(out["jpeg0"]))
    out["jpeg_bare"] = np.frombuffer(bare, dtype=np.uint8)           # decodes to rgb0 with the standard (Annex K) tables
    out["n"] = np.array(len(CASES))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g19_jpeg.npz"), **out)
    print("wrote g19_jpeg.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
