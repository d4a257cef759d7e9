"""The serving worker in front of the REAL native model on an MI355X: the reference's wire format carries the
same reply the drop-in's generate produces, a video source is served through the decoder adaptor, sampling works, and the
streaming-gate endpoint reproduces the reference's streaming trace (golden g6) over HTTP chunks."""
import base64
import io
import json

import numpy as np
import pytest
import torch

from oracle import streammind_oracle as O
from oracle.make_golden import TINY_V as TV, TINY_C as TC, TINY_G as TG, TINY_L as TL
from tests.util_models import build_native, conn_gate_weights

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
PROC = type("P", (), {"crop_size": {"height": TV.image_size, "width": TV.image_size}, "image_mean": list(O.CLIP_MEAN)})()


@pytest.fixture(scope="module")
def worker(tiny_tokenizer):
    from streammind_amd.model import Videollama2MistralForCausalLM
    from streammind_amd.serve.model_worker import ModelWorker
    Wv, Wc, Wl = O.make_vit_weights(TV, 41), conn_gate_weights(TC, TG, 86), O.make_lm_weights(TL, 44)
    m = build_native(TV, TC, TG, Wv, Wc, TL, Wl, max_frames_per_call=8)
    model = Videollama2MistralForCausalLM(m, max_frames=64, max_seq=512, eos_token_id=tiny_tokenizer.eos_token_id)
    return ModelWorker("", "http://w", "t0", True, "x", None, "VideoLLaMA2-7B", loaded=(tiny_tokenizer, model, PROC, 2048))


def _chunks(gen):
    out = list(gen)
    assert all(c.endswith(b"\0") for c in out)
    return [json.loads(c[:-1]) for c in out]


def test_generate_stream_video_source_equals_direct_generate(worker, tmp_path, tiny_tokenizer):
    from streammind_amd.mm_utils import tokenizer_MMODAL_token
    frames = O.synthetic_frames(20, TV.image_size, seed=77, scene_len=4).numpy()
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    prompt = O.initial_prompt()
    req = {"prompt": prompt, "images": [str(path)], "temperature": 0.0, "max_new_tokens": 12, "stop": "</s>"}
    got = _chunks(worker.generate_stream_gate(dict(req)))
    assert all(c["error_code"] == 0 for c in got) and got[-1]["text"].startswith(prompt)
    # the same request through the model API: 8 uniformly spaced frames (model_worker.py:190-195), greedy
    ids = tokenizer_MMODAL_token(prompt, tiny_tokenizer, -201, return_tensors="pt").unsqueeze(0)
    sel = np.linspace(0, 19, 8, dtype=int)
    new = worker.model.generate(ids, images_or_videos=[torch.from_numpy(frames[sel])], modal_list=["video"], do_sample=False, max_new_tokens=12)
    text = tiny_tokenizer.decode(new[0], skip_special_tokens=True)
    assert got[-1]["text"][len(prompt):].strip() == text.strip()
    # base64 pictures: single-frame clips behind <image> sentinels
    buf = io.BytesIO()
    from PIL import Image
    Image.fromarray(frames[3]).save(buf, format="PNG")
    req2 = {"prompt": prompt.replace("<video>", "<image>"), "images": [base64.b64encode(buf.getvalue()).decode()], "temperature": 0.0,
            "max_new_tokens": 6, "stop": "</s>"}
    got2 = _chunks(worker.generate_stream_gate(req2))
    assert got2[-1]["error_code"] == 0
    # sampling (temperature / top_p, model_worker.py:247-251): runs, stays inside the vocabulary, differs between seeds
    req3 = dict(req, temperature=0.9, top_p=0.8, max_new_tokens=10)
    a, b = _chunks(worker.generate_stream_gate(dict(req3)))[-1], _chunks(worker.generate_stream_gate(dict(req3)))[-1]
    assert a["error_code"] == b["error_code"] == 0 and len(a["text"]) >= len(prompt)


def test_stream_frames_endpoint_reproduces_reference_trace(worker, gold):
    from fastapi.testclient import TestClient
    from streammind_amd.serve.model_worker import create_app
    g = gold("g6_stream_tiny")
    n = int(g["n_frames"])
    frames = O.synthetic_frames(n, TV.image_size, seed=int(g["seed_frames"]), scene_len=int(g["scene_len"])).numpy()
    client = TestClient(create_app(worker))
    preds = []
    for i in range(0, n, 4):                                # 4 frames per request, one chunk per frame
        fr = frames[i:i + 4]
        body = client.post("/worker_stream_frames", json={"stream_id": "cam0", "max_new_tokens": int(g["max_new"]),
                                                          "frames": {"u8": base64.b64encode(fr.tobytes()).decode(), "shape": list(fr.shape)}}).content
        for c in [json.loads(x) for x in body.split(b"\0") if x]:
            assert c["error_code"] == 0 and c["stream_id"] == "cam0"
            preds.append(c["cls_pred"])
            assert (c["text"] is not None) == bool(c["cls_pred"])
    assert preds == g["preds"].tolist() and c["frames_seen"] == n
    assert client.post("/worker_get_status").json()["model_names"] == ["VideoLLaMA2-7B"]
    bad = client.post("/worker_stream_frames", json={"stream_id": "cam1", "frames": {"u8": "AAAA", "shape": [1, 9, 9, 3]}}).content
    assert json.loads(bad.split(b"\0")[0])["error_code"] == 1
