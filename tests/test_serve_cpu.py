"""Serving surface against golden g15 = the chunk streams and the registry behaviour recorded from the REFERENCE's
own serve/model_worker.py and serve/controller.py (driven with a stand-in model that streams a fixed reply).  CPU only: the
wire format, error behaviour and the controller logic do not touch the GPU; the worker in front of the real native model is
exercised by tests/test_gpu_serve.py."""
import json

import numpy as np
import pytest
import torch

from streammind_amd.serve.controller import Controller, create_app as controller_app
from streammind_amd.serve.model_worker import ModelWorker, create_app as worker_app


class _Tower:
    num_patches = 576


class FakeModel:
    """generate() feeds fixed ids to the HF streamer: the same stand-in oracle/make_golden.py put behind the reference worker"""
    device = "cpu"

    def __init__(self, ids):
        self.ids = ids
        self.config = type("C", (), {"max_position_embeddings": 2048})()

    def get_vision_tower(self):
        return _Tower()

    def generate(self, **kw):
        st = kw["streamer"]
        st.put(torch.empty(1, 0, dtype=torch.long))
        for t in self.ids:
            st.put(torch.tensor([t]))
        st.end()

    def generate_iter(self, inputs, **kw):          # the native model's chunked form of the same reply (the worker reads this one)
        for i in range(0, len(self.ids), 3):
            yield self.ids[i:i + 3]


PROC = type("P", (), {"crop_size": {"height": 56, "width": 56}, "image_mean": [0.48145466, 0.4578275, 0.40821073]})()


def _worker(tok, reply, **kw):
    ids = tok(reply).input_ids[1:] + [tok.eos_token_id]
    return ModelWorker("", "http://w", "t0", True, "x", None, "VideoLLaMA2-7B", loaded=(tok, FakeModel(ids), PROC, 2048), **kw)


def test_worker_chunk_stream_equals_reference(gold, tiny_tokenizer):
    g = gold("g15_serving")
    w = _worker(tiny_tokenizer, str(g["reply"]))
    reqs, want = json.loads(str(g["requests"])), json.loads(str(g["chunks"]))
    for r, chunks in zip(reqs[:3], want[:3]):          # a reply, a trimmed stop string, an <image>-count mismatch (in-band error)
        got = list(w.generate_stream_gate(dict(r)))
        assert all(c.endswith(b"\0") for c in got)
        assert [c[:-1].decode() for c in got] == chunks
    assert json.loads(want[2][-1])["error_code"] == 1 and json.loads(want[0][-1])["text"].endswith(str(g["reply"]))
    # text-only request: the reference answers with the in-band error (its `modal_token_index` is unbound without images,
    # model_worker.py:262); the drop-in serves it -- the one deliberate difference of the wire behaviour
    assert json.loads(want[3][-1])["error_code"] == 1
    txt = [json.loads(c[:-1]) for c in w.generate_stream_gate(dict(reqs[3]))]
    assert txt[-1]["error_code"] == 0 and txt[-1]["text"].endswith(str(g["reply"]))
    w.model.config.max_position_embeddings = 8
    assert [c[:-1].decode() for c in w.generate_stream_gate(dict(reqs[0]))] == want[4]
    assert w.get_status() == json.loads(str(g["status"]))


def test_worker_http_routes_and_registration(gold, tiny_tokenizer):
    from fastapi.testclient import TestClient
    g = gold("g15_serving")
    ctl = Controller("shortest_queue", start_expiry_thread=False)
    cclient = TestClient(controller_app(ctl))

    def post(url, **kw):                                    # the worker's HTTP calls land in the controller app
        kw.pop("timeout", None)
        return cclient.post(url.replace("http://ctl", ""), **kw)
    ids = tiny_tokenizer(str(g["reply"])).input_ids[1:] + [tiny_tokenizer.eos_token_id]
    w = ModelWorker("http://ctl", "http://w", "t0", False, "x", None, "VideoLLaMA2-7B", loaded=(tiny_tokenizer, FakeModel(ids), PROC, 2048),
                    post=post, start_heart_beat=False)
    assert cclient.post("/list_models").json() == {"models": ["VideoLLaMA2-7B"]}
    assert cclient.post("/get_worker_address", json={"model": "VideoLLaMA2-7B"}).json() == {"address": "http://w"}
    w.send_heart_beat()
    assert ctl.worker_info["http://w"].queue_length == 0
    del ctl.worker_info["http://w"]
    w.send_heart_beat()                                      # {"exist": false} -> the worker registers again (model_worker.py:148-149)
    assert "http://w" in ctl.worker_info
    wclient = TestClient(worker_app(w))
    assert wclient.post("/worker_get_status").json() == {"model_names": ["VideoLLaMA2-7B"], "speed": 1, "queue_length": 0}
    req = json.loads(str(g["requests"]))[0]
    body = wclient.post("/worker_generate_stream", json=req).content
    chunks = [c.decode() for c in body.split(b"\0") if c]
    assert chunks == json.loads(str(g["chunks"]))[0]
    assert w.get_queue_length() == 0                          # the semaphore was released by the background task


def test_controller_registry_equals_reference(gold):
    g = gold("g15_serving")
    want = json.loads(str(g["controller_log"]))
    t = [1000.0]
    c = Controller("shortest_queue", start_expiry_thread=False, clock=lambda: t[0])
    log = []
    for name, q in (("http://w0", 3), ("http://w1", 1), ("http://w2", 2)):
        log.append(c.register_worker(name, True, {"model_names": ["VideoLLaMA2-7B"] if name != "http://w2" else ["other"],
                                                  "speed": 1 + (name == "http://w0"), "queue_length": q}))
    log.append(sorted(c.list_models()))
    log.append([c.get_worker_address("VideoLLaMA2-7B") for _ in range(5)])
    log.append(c.get_worker_address("nope"))
    log.append([c.receive_heart_beat("http://w1", 0), c.receive_heart_beat("http://zz", 0)])
    log.append(c.get_worker_address("VideoLLaMA2-7B"))
    c.worker_info["http://w0"].last_heart_beat -= 31
    c.worker_info["http://w2"].last_heart_beat -= 29
    c.remove_stable_workers_by_expiration()
    log.append(sorted(c.worker_info))
    assert log == want
    assert [x[:-1].decode() for x in c.worker_api_generate_stream({"model": "nope", "prompt": "x"})] == json.loads(str(g["no_worker_chunks"]))
    c2 = Controller("lottery", start_expiry_thread=False)
    for name, sp in (("http://a", 1), ("http://b", 3)):
        c2.register_worker(name, False, {"model_names": ["m"], "speed": sp, "queue_length": 0})
    np.random.seed(int(g["lottery_seed"]))
    assert [c2.get_worker_address("m") for _ in range(12)] == json.loads(str(g["lottery"]))
    with pytest.raises(ValueError):
        Controller("round_robin", start_expiry_thread=False)


def test_controller_proxies_worker_stream(gold, tiny_tokenizer):
    """controller /worker_generate_stream passes the worker's chunks through unchanged"""
    from fastapi.testclient import TestClient
    g = gold("g15_serving")
    w = _worker(tiny_tokenizer, str(g["reply"]))
    wclient = TestClient(worker_app(w))

    class Resp:
        def __init__(self, r):
            self.r, self.status_code = r, r.status_code

        def json(self):
            return self.r.json()

        def iter_lines(self, decode_unicode=False, delimiter=b"\n"):
            return [c for c in self.r.content.split(delimiter)]

    def post(url, **kw):
        kw.pop("timeout", None); kw.pop("stream", None)
        return Resp(wclient.post(url.replace("http://w", ""), **kw))
    c = Controller("shortest_queue", start_expiry_thread=False, post=post)
    assert c.register_worker("http://w", True, None)         # status fetched from the worker itself (controller.py:77-78)
    req = dict(json.loads(str(g["requests"]))[0], model="VideoLLaMA2-7B")
    got = [x[:-1].decode() for x in c.worker_api_generate_stream(req)]
    assert got == json.loads(str(g["chunks"]))[0]
    assert c.worker_api_get_status() == {"model_names": ["VideoLLaMA2-7B"], "speed": 1, "queue_length": 0}


def test_stream_registry_is_bounded(tiny_tokenizer, monkeypatch):
    """/worker_stream_frames keeps device state per client-chosen stream_id: the registry is capped, idle entries are closed first
    (sm_stream_close, not GC), a full registry refuses new ids in-band, {"close": true} releases one (advisor finding, round 2)."""
    import streammind_amd.model.stream_model as SM
    closed = []

    class _S:
        def __init__(self, name):
            self.name = name

        def close(self):
            closed.append(self.name)

    class _M:
        n = 0

        def __init__(self, native, max_frames, max_seq, eos_token_id):
            _M.n += 1
            self.stream = _S(_M.n)
            self.max_frames = max_frames
    monkeypatch.setattr(SM, "Videollama2MistralForCausalLM", _M)
    now = [0.0]
    w = _worker(tiny_tokenizer, "ok", max_streams=2, stream_idle_s=10.0, stream_max_frames=64, clock=lambda: now[0])
    w.model.native, w.model.max_seq = object(), 128
    a = w._stream_state("a", False)
    assert a["model"].max_frames == 64 and w._stream_state("a", False) is a            # same id -> same state
    now[0] = 1.0
    w._stream_state("b", False)
    with pytest.raises(RuntimeError, match="registry full"):
        w._stream_state("c", False)                                                    # both live: refused, nothing evicted
    assert closed == [] and set(w._stream_models) == {"a", "b"}
    out = [json.loads(c[:-1]) for c in w.stream_frames({"stream_id": "c", "frames": []})]
    assert out[-1]["error_code"] == 1                                                  # ... in-band on the wire
    now[0] = 10.5                                                                      # "a" (last used 0.0) is idle now, "b" (1.0) is not
    w._stream_state("c", False)
    assert closed == [1] and set(w._stream_models) == {"b", "c"}
    w._stream_state("b", True)                                                         # reset = close + reopen
    assert closed == [1, 2] and set(w._stream_models) == {"b", "c"}
    out = [json.loads(c[:-1]) for c in w.stream_frames({"stream_id": "c", "close": True})]
    assert out == [{"stream_id": "c", "closed": True, "error_code": 0}] and closed == [1, 2, 3] and set(w._stream_models) == {"b"}
    assert [json.loads(c[:-1]) for c in w.stream_frames({"stream_id": "zz", "close": True})][0]["closed"] is False


def test_a_stalled_client_does_not_hold_the_model_lock(gold, tiny_tokenizer):
    """Advisor (round 4, medium): the chunk stream used to be yielded to the wire INSIDE the model lock.  The lock now belongs to a
    producer thread for as long as the model runs; a reader that stalls after the first chunk (or goes away) holds nothing, and a
    closed response cancels the producer at its next chunk."""
    import time
    g = gold("g15_serving")
    w = _worker(tiny_tokenizer, str(g["reply"]))
    req = json.loads(str(g["requests"]))[0]
    stalled = w.generate_stream_gate(dict(req))
    first = next(stalled)                               # a client that read one chunk and then stopped reading
    assert first.endswith(b"\0")
    deadline = time.time() + 5.0                        # the producer finishes on its own: the lock comes back without the reader
    while w._lock.locked() and time.time() < deadline:
        time.sleep(0.01)
    assert not w._lock.locked()
    other = [c[:-1].decode() for c in w.generate_stream_gate(dict(req))]        # a second request is served in full meanwhile
    assert other == json.loads(str(g["chunks"]))[0]
    stalled.close()                                     # the client disconnects: GeneratorExit, nothing left behind
    assert not w._lock.locked()
    # a producer that is still decoding when the client leaves stops at its next chunk
    seen = []

    class Slow(FakeModel):
        def generate_iter(self, inputs, **kw):
            for i in range(0, len(self.ids), 3):
                seen.append(i)
                time.sleep(0.02)
                yield self.ids[i:i + 3]

    w2 = _worker(tiny_tokenizer, str(g["reply"]))
    w2.model = Slow(w2.model.ids)
    gen = w2.generate_stream_gate(dict(req))
    next(gen)
    gen.close()
    time.sleep(0.3)
    assert not w2._lock.locked() and len(seen) < len(range(0, len(w2.model.ids), 3))
