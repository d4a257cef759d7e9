"""Worker registry and dispatcher: mirror of streammind/serve/controller.py:28-298.

Routes (all POST, JSON): /register_worker {worker_name, check_heart_beat, worker_status?}; /refresh_all_workers;
/list_models -> {models}; /get_worker_address {model} -> {address}; /receive_heart_beat {worker_name, queue_length} -> {exist};
/worker_generate_stream (proxy: the worker's `\\0`-delimited chunks are passed through; error_code 2 = no worker,
3 = worker timeout); /worker_get_status (sum over the workers).  Workers that registered with check_heart_beat expire
CONTROLLER_HEART_BEAT_EXPIRATION seconds after their last beat."""
import argparse
import dataclasses
import json
import threading
import time
from enum import Enum, auto
from typing import Callable, List, Optional

import numpy as np

from ..constants import CONTROLLER_HEART_BEAT_EXPIRATION
from ..utils import server_error_msg


class DispatchMethod(Enum):
    LOTTERY = auto()
    SHORTEST_QUEUE = auto()

    @classmethod
    def from_str(cls, name: str) -> "DispatchMethod":
        if name == "lottery":
            return cls.LOTTERY
        if name == "shortest_queue":
            return cls.SHORTEST_QUEUE
        raise ValueError("Invalid dispatch method")


@dataclasses.dataclass
class WorkerInfo:
    model_names: List[str]
    speed: int
    queue_length: int
    check_heart_beat: bool
    last_heart_beat: float


def _http_post(url: str, **kw):
    import requests
    return requests.post(url, **kw)


class Controller:
    def __init__(self, dispatch_method: str = "shortest_queue", start_expiry_thread: bool = True,
                 post: Callable = _http_post, clock: Callable[[], float] = time.time):
        self.worker_info = {}
        self.dispatch_method = DispatchMethod.from_str(dispatch_method)
        self._post, self._clock = post, clock
        if start_expiry_thread:
            self.heart_beat_thread = threading.Thread(target=self._expiry_loop, daemon=True)
            self.heart_beat_thread.start()

    def _expiry_loop(self):
        while True:
            time.sleep(CONTROLLER_HEART_BEAT_EXPIRATION)
            self.remove_stable_workers_by_expiration()

    def register_worker(self, worker_name: str, check_heart_beat: bool, worker_status: Optional[dict]) -> bool:
        if not worker_status:
            worker_status = self.get_worker_status(worker_name)
        if not worker_status:
            return False
        self.worker_info[worker_name] = WorkerInfo(worker_status["model_names"], worker_status["speed"], worker_status["queue_length"],
                                                   check_heart_beat, self._clock())
        return True

    def get_worker_status(self, worker_name: str) -> Optional[dict]:
        try:
            r = self._post(worker_name + "/worker_get_status", timeout=5)
        except Exception:
            return None
        if r.status_code != 200:
            return None
        return r.json()

    def remove_worker(self, worker_name: str) -> None:
        del self.worker_info[worker_name]

    def refresh_all_workers(self) -> None:
        old = dict(self.worker_info)
        self.worker_info = {}
        for name, info in old.items():
            self.register_worker(name, info.check_heart_beat, None)        # workers that no longer answer are dropped

    def list_models(self) -> List[str]:
        names = set()
        for info in self.worker_info.values():
            names.update(info.model_names)
        return list(names)

    def get_worker_address(self, model_name: str) -> str:
        cand = [(n, i) for n, i in self.worker_info.items() if model_name in i.model_names]
        if self.dispatch_method == DispatchMethod.LOTTERY:
            speeds = np.array([i.speed for _, i in cand], dtype=np.float32)
            norm = np.sum(speeds)
            if norm < 1e-4:
                return ""
            return cand[np.random.choice(np.arange(len(cand)), p=speeds / norm)][0]       # speed-weighted draw
        if not cand:
            return ""
        k = int(np.argmin([i.queue_length / i.speed for _, i in cand]))
        cand[k][1].queue_length += 1                                                       # optimistic until its next heart beat
        return cand[k][0]

    def receive_heart_beat(self, worker_name: str, queue_length: int) -> bool:
        if worker_name not in self.worker_info:
            return False
        self.worker_info[worker_name].queue_length = queue_length
        self.worker_info[worker_name].last_heart_beat = self._clock()
        return True

    def remove_stable_workers_by_expiration(self) -> None:
        expire = self._clock() - CONTROLLER_HEART_BEAT_EXPIRATION
        for name in [n for n, i in self.worker_info.items() if i.check_heart_beat and i.last_heart_beat < expire]:
            self.remove_worker(name)

    def worker_api_generate_stream(self, params: dict):
        addr = self.get_worker_address(params["model"])
        if not addr:
            # controller.py:218-225 does not return here: its request to "" + "/worker_generate_stream" fails at once and the
            # client receives the error-3 chunk right behind this one -- kept, the byte stream is the contract
            yield json.dumps({"text": server_error_msg, "error_code": 2}).encode() + b"\0"
        try:
            if not addr:
                raise ConnectionError("no worker address")
            response = self._post(addr + "/worker_generate_stream", json=params, stream=True, timeout=5)
            for chunk in response.iter_lines(decode_unicode=False, delimiter=b"\0"):
                if chunk:
                    yield chunk + b"\0"
        except Exception:
            yield json.dumps({"text": server_error_msg, "error_code": 3}).encode() + b"\0"

    def worker_api_get_status(self) -> dict:
        names, speed, qlen = set(), 0, 0
        for name in self.worker_info:
            st = self.get_worker_status(name)
            if st is not None:
                names.update(st["model_names"])
                speed += st["speed"]
                qlen += st["queue_length"]
        return {"model_names": list(names), "speed": speed, "queue_length": qlen}


def create_app(controller: Controller):
    from fastapi import FastAPI, Request
    from fastapi.responses import StreamingResponse
    app = FastAPI()

    @app.post("/register_worker")
    async def register_worker(request: Request):
        data = await request.json()
        controller.register_worker(data["worker_name"], data["check_heart_beat"], data.get("worker_status", None))

    @app.post("/refresh_all_workers")
    async def refresh_all_workers():
        controller.refresh_all_workers()

    @app.post("/list_models")
    async def list_models():
        return {"models": controller.list_models()}

    @app.post("/get_worker_address")
    async def get_worker_address(request: Request):
        data = await request.json()
        return {"address": controller.get_worker_address(data["model"])}

    @app.post("/receive_heart_beat")
    async def receive_heart_beat(request: Request):
        data = await request.json()
        return {"exist": controller.receive_heart_beat(data["worker_name"], data["queue_length"])}

    @app.post("/worker_generate_stream")
    async def worker_api_generate_stream(request: Request):
        params = await request.json()
        return StreamingResponse(controller.worker_api_generate_stream(params))

    @app.post("/worker_get_status")
    async def worker_api_get_status(request: Request):
        return controller.worker_api_get_status()

    return app


if __name__ == "__main__":
    import uvicorn
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", type=str, default="localhost")
    ap.add_argument("--port", type=int, default=21001)
    ap.add_argument("--dispatch-method", type=str, choices=["lottery", "shortest_queue"], default="shortest_queue")
    a = ap.parse_args()
    uvicorn.run(create_app(Controller(a.dispatch_method)), host=a.host, port=a.port, log_level="info")
