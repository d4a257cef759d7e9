"""Model worker: mirror of streammind/serve/model_worker.py:85-397 in front of the native model.

  POST /worker_generate_stream  {prompt, images?: [base64 image | video path], temperature, top_p, max_new_tokens, stop}
        -> a stream of  json({"text": <prompt + text so far>, "error_code": 0}) + b"\\0"   (model_worker.py:286-300);
        errors are in-band: error_code 1 with server_error_msg for ValueError / device errors / anything else (:302-332)
  POST /worker_get_status       -> {"model_names": [...], "speed": 1, "queue_length": n}
  POST /worker_stream_frames    (NOT in the reference, which never serves its own streaming path -- SURVEY 3.5): one tick of a
        named stream: {stream_id, frames: [base64 image, ...] | {"u8": base64, "shape": [n,H,W,3]}, max_new_tokens?, reset?}
        -> chunks json({"stream_id", "frames_seen", "cls_pred", "text": reply | null, "error_code"}) + b"\\0"

Registration / heart beat towards the controller as model_worker.py:62-66,117-149.  Differences forced by the path: `images`
given as base64 pictures are single-frame clips (the Mamba connector has no separate image branch); a video path is opened
through video_io.open_video (decoder adaptor) and sampled at 8 uniformly spaced frames like :190-195."""
import argparse
import asyncio
import base64
import json
import contextlib
import threading
import time
import uuid
from functools import partial
from io import BytesIO
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from ..constants import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN, DEFAULT_VIDEO_TOKEN, MMODAL_TOKEN_INDEX,
                         WORKER_HEART_BEAT_INTERVAL)
from ..mm_utils import KeywordsStoppingCriteria, process_video, tokenizer_MMODAL_token
from ..utils import pretty_print_semaphore, server_error_msg

STREAM_CHECK_MULTIPLE = 20


class _WordBoundaryDetokenizer:
    """ids in, text out, one id at a time: the text decoded so far is released up to the last space (whole words only; everything on a
    newline or a CJK character, the rest at the end) -- the release rule of the HF text streamer the reference worker reads from
    (serve/model_worker.py:262-288), so the chunk boundaries on the wire are the reference's for the same ids."""

    def __init__(self, tokenizer):
        self.tok, self.ids, self.sent = tokenizer, [], 0

    @staticmethod
    def _cjk(cp: int) -> bool:
        return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F or
                0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)

    def put(self, tok_id: int) -> str:
        self.ids.append(tok_id)
        text = self.tok.decode(self.ids, skip_special_tokens=True)
        if text.endswith("\n"):
            piece, self.ids, self.sent = text[self.sent:], [], 0
        elif text and self._cjk(ord(text[-1])):
            piece = text[self.sent:]
            self.sent += len(piece)
        else:
            piece = text[self.sent:text.rfind(" ") + 1]
            self.sent += len(piece)
        return piece

    def end(self) -> str:
        piece = self.tok.decode(self.ids, skip_special_tokens=True)[self.sent:] if self.ids else ""
        self.ids, self.sent = [], 0
        return piece


def load_image_from_base64(image: str):
    from PIL import Image
    return Image.open(BytesIO(base64.b64decode(image)))


def _http_post(url: str, **kw):
    import requests
    return requests.post(url, **kw)


class ModelWorker:
    def __init__(self, controller_addr, worker_addr, worker_id, no_register, model_path, model_base, model_name, load_8bit=False,
                 load_4bit=False, device="cuda", *, loaded=None, post: Callable = _http_post, limit_model_concurrency: int = 5,
                 start_heart_beat: bool = True, keywords=(), max_streams: int = 16, stream_idle_s: float = 600.0,
                 stream_max_frames: int = 4096, clock: Callable = time.monotonic):
        self.controller_addr, self.worker_addr, self.worker_id = controller_addr, worker_addr, worker_id
        self.model_path = model_path
        if model_name is None:
            from ..mm_utils import get_model_name_from_path
            model_name = get_model_name_from_path(model_path)
        self.model_name, self.device = model_name, device
        self._post = post
        self.limit_model_concurrency = limit_model_concurrency
        self.model_semaphore: Optional[asyncio.Semaphore] = None
        self.global_counter = 0
        self.keywords = [k for k in keywords if k]                 # assets/keywords.txt of the reference (:45-55); empty by default
        if loaded is None:
            from ..model.builder import load_pretrained_model
            loaded = load_pretrained_model(model_path, model_base, self.model_name, load_8bit, load_4bit, device=device)
        self.tokenizer, self.model, self.image_processor, self.context_len = loaded
        self.is_multimodal = "videollama2" in self.model_name.lower() or "vlb" in self.model_name.lower()
        # /worker_stream_frames: stream_id -> per-stream state.  Every entry owns device memory (KV cache, token store, prefill
        # workspaces: ~1 GB at Mistral-7B sizes), and the ids come from the network: the registry is bounded -- at most
        # `max_streams` entries, entries idle for `stream_idle_s` are closed first, a request that would exceed the cap while
        # every entry is live is refused in-band (error_code 1), and {"close": true} releases an entry explicitly.
        self._stream_models: Dict[str, dict] = {}
        self.max_streams, self.stream_idle_s, self.stream_max_frames, self._clock = int(max_streams), float(stream_idle_s), int(stream_max_frames), clock
        self._lock = threading.Lock()                               # one generation at a time per model object (it holds stream state)
        if not no_register:
            self.register_to_controller()
            if start_heart_beat:
                self.heart_beat_thread = threading.Thread(target=self._heart_beat_loop, daemon=True)
                self.heart_beat_thread.start()

    # ---- controller protocol (model_worker.py:117-163)
    def _heart_beat_loop(self):
        while True:
            time.sleep(WORKER_HEART_BEAT_INTERVAL)
            self.send_heart_beat()

    def register_to_controller(self):
        r = self._post(self.controller_addr + "/register_worker",
                       json={"worker_name": self.worker_addr, "check_heart_beat": True, "worker_status": self.get_status()})
        assert r.status_code == 200

    def send_heart_beat(self):
        while True:
            try:
                ret = self._post(self.controller_addr + "/receive_heart_beat",
                                 json={"worker_name": self.worker_addr, "queue_length": self.get_queue_length()}, timeout=5)
                exist = ret.json()["exist"]
                break
            except Exception:
                time.sleep(5)
        if not exist:
            self.register_to_controller()

    def get_queue_length(self) -> int:
        sem = self.model_semaphore
        if sem is None:
            return 0
        return self.limit_model_concurrency - sem._value + (len(sem._waiters) if sem._waiters is not None else 0)

    def get_status(self) -> dict:
        return {"model_names": [self.model_name], "speed": 1, "queue_length": self.get_queue_length()}

    # ---- keyword screen (model_worker.py:69-83)
    def safety_check(self, text: str) -> Optional[str]:
        if self.keywords and any(x in text.lower() for x in self.keywords):
            return ("The output contains political, erotic and other unsafe content that violates local laws. "
                    "Please re-enter your question.")
        return None

    def input_safety_check(self, text: str) -> Optional[str]:
        if self.keywords and any(x in text.lower() for x in self.keywords):
            return ("Your input question contains political, erotic and other unsafe content that violates local laws. "
                    "Please re-enter your question.")
        return None

    # ---- /worker_generate_stream (model_worker.py:165-300)
    def _load_clips(self, images_or_videos, prompt: str):
        """-> (clips, modal_list, replace_token, modal_token_index)"""
        if len(images_or_videos) != prompt.count(DEFAULT_IMAGE_TOKEN) and len(images_or_videos) != prompt.count(DEFAULT_VIDEO_TOKEN):
            raise ValueError("Number of images/videos does not match number of <image>/<video> tokens in prompt")
        cfg = self.model.config
        ar = getattr(cfg, "image_aspect_ratio", None)
        try:
            pics = [np.asarray(load_image_from_base64(im).convert("RGB")) for im in images_or_videos]
        except Exception:                                   # model_worker.py:186-188: not base64 pictures -> a video path
            pics = None
        if pics is not None:
            clips = [process_video([p], self.image_processor, aspect_ratio=ar, num_frames=1) for p in pics]
            return clips, ["image"] * len(clips), DEFAULT_IMAGE_TOKEN, MMODAL_TOKEN_INDEX["IMAGE"]
        else:
            from ..video_io import open_video
            vr = open_video(images_or_videos[0])
            ids = np.linspace(0, len(vr) - 1, 8, dtype=int)
            frames = np.asarray(vr.get_batch(ids).asnumpy())
            clip = process_video(frames, self.image_processor, aspect_ratio=ar, num_frames=len(frames))
            return [clip], ["video"], DEFAULT_VIDEO_TOKEN, MMODAL_TOKEN_INDEX["VIDEO"]

    # ---- the model lock never waits for a client
    def _produce_locked(self, body):
        """Run `body(put, cancelled)` on a producer thread that holds the model lock for exactly as long as the MODEL runs, and hand what
        it `put`s to the calling generator through a queue: the HTTP response is written outside the lock, so a slow, stalled or gone
        client holds nothing (round 4 yielded to the wire inside `with self._lock`: one stuck reader blocked every other request; a
        closed generator now sets `cancelled`, which the body polls between decode chunks / ticks).  The queue is bounded by the
        request itself (<= 1024 new tokens, <= the frames posted).  Exceptions of the body re-raise in the consumer."""
        import queue
        q, cancelled, DONE = queue.Queue(), threading.Event(), object()

        dev = getattr(self.model, "device", None)

        def run():
            try:
                # a fresh thread starts on device 0 with device 0's current stream: pin it to the model's device (round-5 advisor; `--device cuda:N`)
                ctx = torch.cuda.device(dev) if (dev is not None and torch.cuda.is_available() and getattr(dev, "type", "cuda") == "cuda") else contextlib.nullcontext()
                with ctx, self._lock, torch.inference_mode():
                    body(q.put, cancelled.is_set)
                q.put(DONE)
            except BaseException as e:                      # delivered in-band to the consumer
                q.put(e)

        t = threading.Thread(target=run, name="streammind-model-producer", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is DONE:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            cancelled.set()                                 # GeneratorExit (client went away) or normal end: the producer stops at its next check

    @torch.inference_mode()
    def generate_stream(self, params: dict):
        tokenizer, model = self.tokenizer, self.model
        ori_prompt = prompt = params["prompt"]
        media = params.get("images", None)
        clip_kw, n_visual, modal_index = {}, 0, MMODAL_TOKEN_INDEX["IMAGE"]
        if media is not None and len(media) and self.is_multimodal:
            clips, modal_list, placeholder, modal_index = self._load_clips(media, prompt)
            if getattr(model.config, "mm_use_im_start_end", False):
                placeholder = DEFAULT_IM_START_TOKEN + placeholder + DEFAULT_IM_END_TOKEN
            prompt = prompt.replace(DEFAULT_IMAGE_TOKEN, placeholder)
            n_visual = prompt.count(placeholder) * model.get_vision_tower().num_patches
            clip_kw = {"images_or_videos": clips, "modal_list": modal_list}
        temperature, top_p = float(params.get("temperature", 1.0)), float(params.get("top_p", 1.0))
        stop_str = params.get("stop", None)
        input_ids = tokenizer_MMODAL_token(prompt, tokenizer, modal_index, return_tensors="pt").unsqueeze(0)
        budget = min(int(params.get("max_new_tokens", 256)), 1024,
                     getattr(model.config, "max_position_embeddings", 2048) - input_ids.shape[-1] - n_visual)
        if budget < 1:
            yield json.dumps({"text": ori_prompt + "Exceeds max token length. Please start a new conversation, thanks.", "error_code": 0}).encode() + b"\0"
            return
        stops = [KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)]
        # The native decode loop IS the stream: every accepted id (decode_chunk greedy steps per host sync) goes through an incremental
        # detokeniser and out on the wire from this very generator -- no generate() thread, no streamer queue.  The wire format is the
        # reference's: one NUL-terminated JSON object per new id with the whole text so far (text is released at word boundaries, as the
        # reference's streamer releases it), the moderation hook every STREAM_CHECK_MULTIPLE tokens, a trailing stop string trimmed.
        detok = _WordBoundaryDetokenizer(tokenizer)
        text, since_check = ori_prompt, 0

        def emit(piece):
            nonlocal text, since_check
            text += piece
            since_check += len(tokenizer.encode(piece))
            if since_check >= STREAM_CHECK_MULTIPLE:
                since_check = 0
                msg = self.safety_check(text)
                if msg:
                    return json.dumps({"text": msg, "error_code": 1}).encode() + b"\0", True
            if stop_str and text.endswith(stop_str):
                text = text[:-len(stop_str)]
            return json.dumps({"text": text, "error_code": 0}).encode() + b"\0", False

        def decode(put, cancelled):                          # under the model lock, on the producer thread: ids only
            for chunk in model.generate_iter(input_ids, max_new_tokens=budget, stopping_criteria=stops, do_sample=temperature > 0.001,
                                             temperature=temperature, top_p=top_p, **clip_kw):
                put([int(t) for t in chunk])
                if cancelled():
                    return

        for chunk in self._produce_locked(decode):              # detokenise, moderate and write to the wire OUTSIDE the lock
            for tok_id in chunk:
                out, blocked = emit(detok.put(tok_id))
                yield out
                if blocked:
                    return
        out, _ = emit(detok.end())
        yield out

    def generate_stream_gate(self, params: dict):
        try:
            msg = self.input_safety_check(params.get("prompt", ""))
            if msg:
                yield json.dumps({"text": msg, "error_code": 1}).encode() + b"\0"
                return
            for x in self.generate_stream(params):
                yield x
        except Exception as e:                              # ValueError, device errors, anything: all in-band, error_code 1 (:312-332)
            print("Caught", type(e).__name__, e)
            yield json.dumps({"text": server_error_msg, "error_code": 1}).encode() + b"\0"

    # ---- /worker_stream_frames: the streaming gate behind HTTP (no reference counterpart)
    def _close_stream(self, stream_id: str) -> bool:
        st = self._stream_models.pop(stream_id, None)
        if st is None:
            return False
        st["model"].stream.close()                          # sm_stream_close: the device buffers go back now, not at GC time
        return True

    def _evict_idle_streams(self) -> None:
        now = self._clock()
        for sid in [k for k, v in self._stream_models.items() if now - v["last_used"] > self.stream_idle_s]:
            self._close_stream(sid)

    def _stream_state(self, stream_id: str, reset: bool):
        from ..model.stream_model import Videollama2MistralForCausalLM
        if reset:
            self._close_stream(stream_id)
        st = self._stream_models.get(stream_id)
        if st is None:
            self._evict_idle_streams()
            if len(self._stream_models) >= self.max_streams:
                raise RuntimeError(f"stream registry full ({self.max_streams} live streams)")
            m = Videollama2MistralForCausalLM(self.model.native, max_frames=self.stream_max_frames,
                                              max_seq=self.model.max_seq, eos_token_id=self.tokenizer.eos_token_id)
            st = {"model": m, "prompt": None}
            self._stream_models[stream_id] = st
        st["last_used"] = self._clock()
        return st

    @torch.inference_mode()
    def stream_frames(self, params: dict):
        """one tick per decoded frame, exactly the demo loop (eval/video_score_stream_demo.py:283-299) on a named stream"""
        from ..eval.video_score_stream_demo import infer
        try:
            sid = str(params["stream_id"])
            if params.get("close", False):                  # explicit release of a stream's device state
                with self._lock:
                    closed = self._close_stream(sid)
                yield json.dumps({"stream_id": sid, "closed": closed, "error_code": 0}).encode() + b"\0"
                return
            fr = params["frames"]                         # decoded BEFORE the registry is touched: lookup / create and every tick of this request
            if isinstance(fr, dict):                      # are then ONE critical section (a close / reset / eviction cannot land in between)
                frames = np.frombuffer(base64.b64decode(fr["u8"]), dtype=np.uint8).reshape(fr["shape"])
            else:
                frames = np.stack([np.asarray(load_image_from_base64(f).convert("RGB")) for f in fr])
            ar = getattr(self.model.config, "image_aspect_ratio", None)
            def ticks(put, cancelled):                       # lookup / create and every tick of this request are ONE critical section
                st = self._stream_state(sid, bool(params.get("reset", False)))
                for i in range(len(frames)):
                    if cancelled():
                        return
                    video = process_video(frames[i:i + 1], self.image_processor, aspect_ratio=ar, num_frames=1)
                    text, st["prompt"] = infer(st["model"], video, "", self.tokenizer, prompt=st["prompt"],
                                               max_new_tokens=int(params.get("max_new_tokens", 1024)))
                    st["last_used"] = self._clock()
                    put(json.dumps({"stream_id": sid, "frames_seen": st["model"].stream.num_frames, "cls_pred": int(text is not None),
                                    "text": text, "error_code": 0}).encode() + b"\0")

            yield from self._produce_locked(ticks)
        except Exception as e:
            print("Caught", type(e).__name__, e)
            yield json.dumps({"text": server_error_msg, "error_code": 1}).encode() + b"\0"


def create_app(worker: ModelWorker):
    from fastapi import BackgroundTasks, FastAPI, Request
    from fastapi.responses import StreamingResponse
    app = FastAPI()

    def release(fn=None):
        worker.model_semaphore.release()
        if fn is not None:
            fn()

    async def _stream(request: Request, gen_fn):
        worker.global_counter += 1
        params = await request.json()
        if worker.model_semaphore is None:
            worker.model_semaphore = asyncio.Semaphore(worker.limit_model_concurrency)
        await worker.model_semaphore.acquire()
        beat = worker.send_heart_beat if worker.controller_addr else None
        if beat:
            beat()
        background = BackgroundTasks()
        background.add_task(partial(release, fn=beat))
        return StreamingResponse(gen_fn(params), background=background)

    @app.post("/worker_generate_stream")
    async def generate_stream(request: Request):
        return await _stream(request, worker.generate_stream_gate)

    @app.post("/worker_stream_frames")
    async def stream_frames(request: Request):
        return await _stream(request, worker.stream_frames)

    @app.post("/worker_get_status")
    async def get_status(request: Request):
        return worker.get_status()

    return app


if __name__ == "__main__":
    import uvicorn
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", type=str, default="localhost")
    ap.add_argument("--port", type=int, default=21002)
    ap.add_argument("--worker-address", type=str, default="http://localhost:21002")
    ap.add_argument("--controller-address", type=str, default="http://localhost:21001")
    ap.add_argument("--model-path", type=str, required=True)
    ap.add_argument("--model-base", type=str, default=None)
    ap.add_argument("--model-name", type=str)
    ap.add_argument("--device", type=str, default="cuda")
    ap.add_argument("--limit-model-concurrency", type=int, default=5)
    ap.add_argument("--no-register", action="store_true")
    ap.add_argument("--load-8bit", action="store_true")
    ap.add_argument("--load-4bit", action="store_true")
    a = ap.parse_args()
    w = ModelWorker(a.controller_address, a.worker_address, str(uuid.uuid4())[:6], a.no_register, a.model_path, a.model_base, a.model_name,
                    a.load_8bit, a.load_4bit, a.device, limit_model_concurrency=a.limit_model_concurrency)
    uvicorn.run(create_app(w), host=a.host, port=a.port, log_level="info")
