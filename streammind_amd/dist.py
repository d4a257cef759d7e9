"""Multi-GPU plumbing of the streaming path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on
ROCm, "gloo" on CPU for the tests).

The reference's inference path issues NO collective per frame or token: ranks own whole streams, split
contiguously (`EvalDistributedSampler`, eval/inference_video_score_stream_ddp.py:191-213).  The one exchange step the
north-star adds is a variable-size all-gather of the frame tokens of streams whose gate fired, following the reference's own
two-phase pattern `allgather_diff_shape` (streammind/dist.py:122-146): sizes first, then the padded payload.

Two forms:
  * `allgather_gated_tokens`  -- the blocking two-phase exchange (sizes, host read, payload): simple, one host sync per call.
  * `GatedTokenExchange`      -- the form the streaming loop uses: ranks fire on DIFFERENT ticks, so some word has to tell a
    rank that a peer fired; that word is a 4-byte-per-rank count all-gather issued ASYNCHRONOUSLY on a side stream every tick
    and read back one tick later from pinned memory (its event is long complete: no host stall, no bubble in the compute
    stream).  The payload collective is launched only for ticks whose counts say that somebody fired; a silent tick moves no
    payload and blocks nobody.

Also here: the thin process-group helpers callers of the reference's `videollama2.dist` use (streammind/dist.py:16-119,
149-215), same names and degradation rules (everything is a no-op in a single process)."""
from __future__ import annotations

import datetime
import functools
import os
import sys
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
import torch.distributed as tdist

_state = {"rank": 0, "local_rank": 0, "world": 1, "device": "cpu", "initialized": False}


# ------------------------------------------------------------------------------------------------ reference-named helpers
def initialized() -> bool:
    return _state["initialized"]


def initialize(fork=False, backend="nccl", gpu_id_if_not_distibuted=0, timeout=30):
    """streammind/dist.py:20-49: RANK unset -> single process on one GPU; else one rank per GPU, init_process_group."""
    if not torch.cuda.is_available():
        print("[dist initialize] cuda is not available, use cpu instead", file=sys.stderr)
        return
    if "RANK" not in os.environ:
        torch.cuda.set_device(gpu_id_if_not_distibuted)
        _state["device"] = torch.device("cuda", torch.cuda.current_device())
        print(f'[dist initialize] env variable "RANK" is not set, use {_state["device"]} as the device', file=sys.stderr)
        return
    global_rank, num_gpus = int(os.environ["RANK"]), torch.cuda.device_count()
    local_rank = global_rank % num_gpus
    torch.cuda.set_device(local_rank)
    if not tdist.is_initialized():
        tdist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout * 60))
    _state.update(rank=tdist.get_rank(), local_rank=local_rank, world=tdist.get_world_size(),
                  device=torch.device("cuda", local_rank), initialized=True)
    print(f"[lrk={get_local_rank()}, rk={get_rank()}]")


def get_rank() -> int:
    return _state["rank"]


def get_local_rank() -> int:
    return _state["local_rank"]


def get_world_size() -> int:
    return _state["world"]


def get_device():
    return _state["device"]


def is_master() -> bool:
    return _state["rank"] == 0


def is_local_master() -> bool:
    return _state["local_rank"] == 0


def new_group(ranks: List[int]):
    """dist.py:86-89"""
    if _state["initialized"]:
        return tdist.new_group(ranks=ranks)
    return None


def barrier():
    if _state["initialized"]:
        tdist.barrier()


def _comm_device(t: torch.Tensor) -> torch.Tensor:
    return t if (t.is_cuda or tdist.get_backend() != "nccl") else t.cuda()


def allreduce(t: torch.Tensor, async_op=False):
    if not _state["initialized"]:
        return None
    cu = _comm_device(t.detach())
    if cu.device == t.device:                 # shares t's storage: reduced in place, nothing to copy back (and async stays async)
        return tdist.all_reduce(cu, async_op=async_op)
    # a CPU tensor under nccl went through a device copy (reference dist.py:97-109): the copy-back needs the reduced values,
    # so the collective is completed here whatever async_op says
    tdist.all_reduce(cu)
    t.copy_(cu.cpu())
    return None


def allgather(t: torch.Tensor, cat=True) -> Union[List[torch.Tensor], torch.Tensor]:
    if _state["initialized"]:
        t = _comm_device(t)
        ls = [torch.empty_like(t) for _ in range(_state["world"])]
        tdist.all_gather(ls, t)
    else:
        ls = [t]
    return torch.cat(ls, dim=0) if cat else ls


def allgather_diff_shape(t: torch.Tensor, cat=True) -> Union[List[torch.Tensor], torch.Tensor]:
    """streammind/dist.py:122-146: sizes first, pad dim 0 to the maximum, gather, trim"""
    if _state["initialized"]:
        ls = allgather_gated_tokens(_comm_device(t), None) or [t.new_empty((0, *t.shape[1:])) for _ in range(_state["world"])]
    else:
        ls = [t]
    return torch.cat(ls, dim=0) if cat else ls


def broadcast(t: torch.Tensor, src_rank) -> None:
    if _state["initialized"]:
        cu = _comm_device(t.detach())
        tdist.broadcast(cu, src=src_rank)
        if cu.device != t.device:             # only a CPU tensor that travelled through a device copy is copied back
            t.copy_(cu.cpu())


def master_only(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        force = kwargs.pop("force", False)
        ret = func(*args, **kwargs) if (force or is_master()) else None
        barrier()
        return ret
    return wrapper


def finalize():
    if _state["initialized"]:
        tdist.destroy_process_group()
        _state["initialized"] = False


# ------------------------------------------------------------------------------------------------ stream partition
def partition_streams(n_streams: int, world: int, rank: int) -> Tuple[int, int]:
    """[beg, end) of the streams rank `rank` owns: np.linspace blocks, exactly EvalDistributedSampler's split."""
    seps = np.linspace(0, n_streams, world + 1, dtype=int)
    return int(seps[rank]), int(seps[rank + 1])


# ------------------------------------------------------------------------------------------------ gated-token exchange
def _default_device(group) -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if tdist.get_backend(group) == "nccl" else torch.device("cpu")


def allgather_gated_tokens(tokens: Optional[torch.Tensor], d_model: Optional[int], group=None) -> Optional[List[torch.Tensor]]:
    """Blocking form.  tokens: [n_fired, ...] of THIS rank (None / 0 rows when its gate stayed silent; then d_model gives the
    row width).  Returns None when no rank fired (payload collective skipped), else the list of per-rank tensors.

    Phase 1: all-gather of one int32 count per rank (latency-bound, 4 B x world) and a host read of it.  Phase 2 (only if
    max > 0): all-gather of the payload padded to the max count.  Messages are <= ~1 MB on an 8-GPU xGMI node, far below the
    per-link bandwidth regime: RCCL's direct all-gather is one hop per peer."""
    world = tdist.get_world_size(group)
    dev = tokens.device if tokens is not None else _default_device(group)
    n = 0 if tokens is None else int(tokens.shape[0])
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    counts = torch.empty(world, dtype=torch.int32, device=dev)
    tdist.all_gather_into_tensor(counts, cnt, group=group)
    counts = counts.tolist()
    return _payload_allgather(tokens, counts, d_model, dev, group)


def _payload_allgather(tokens, counts, d_model, dev, group):
    mx = max(counts)
    if mx == 0:
        return None
    world = len(counts)
    n = 0 if tokens is None else int(tokens.shape[0])
    tail = tuple(tokens.shape[1:]) if tokens is not None else (d_model,)
    dtype = tokens.dtype if tokens is not None else torch.float32
    pad = torch.zeros((mx, *tail), dtype=dtype, device=dev)
    if n:
        pad[:n] = tokens
    out = torch.empty((world * mx, *tail), dtype=dtype, device=dev)
    tdist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * mx: r * mx + counts[r]] for r in range(world)]


class GatedTokenExchange:
    """Pipelined exchange for ranks that fire on different ticks.

        ex = GatedTokenExchange(d_model)
        for every tick:                      # same number of ticks on every rank (one per perceive call / frame)
            prev = ex.tick(tokens_or_None)   # result of the PREVIOUS tick: None (nobody fired) or [per-rank tensors]
        last = ex.flush()                    # result of the final tick

    Per tick the only traffic is the asynchronous 4-byte-per-rank count all-gather (on a side stream under nccl); its host
    read happens one tick later from pinned memory.  `payload_collectives` counts the ticks that moved tokens,
    `host_waits_us` what the deferred reads cost (~0 when a tick of compute sits between post and read)."""

    def __init__(self, d_model: int, group=None, dtype: torch.dtype = torch.float32, device: Optional[torch.device] = None):
        self.group, self.d_model, self.dtype = group, d_model, dtype
        self.world = tdist.get_world_size(group)
        self.dev = device or _default_device(group)
        self.cuda = self.dev.type == "cuda"
        self.side = torch.cuda.Stream(self.dev) if self.cuda else None
        self._pending = None          # (counts host tensor, ready event | work handle, this rank's tokens of that tick)
        self._bufs = [None, None]
        self.ticks = self.payload_collectives = 0
        self.host_wait_s = 0.0

    def _buffers(self, par: int):
        """per tick parity, allocated ONCE: pinned count word, device count word + gathered counts, pinned host mirror, event -- the
        hot tick allocates and pins nothing (round 4 built and pinned three tensors per tick on what may be the only path a real
        8-GPU run takes).  Two parities: the mirror of tick t is read while tick t + 1 is already posted."""
        if self._bufs[par] is None:
            self._bufs[par] = (torch.empty(1, dtype=torch.int32).pin_memory(), torch.empty(1, dtype=torch.int32, device=self.dev),
                               torch.empty(self.world, dtype=torch.int32, device=self.dev), torch.empty(self.world, dtype=torch.int32).pin_memory(),
                               torch.cuda.Event())
        return self._bufs[par]

    def _post(self, tokens: Optional[torch.Tensor]):
        n = 0 if tokens is None else int(tokens.shape[0])
        if self.cuda:
            cnt_h, cnt, counts, host, ev = self._buffers(self.ticks & 1)
            cnt_h[0] = n                          # this parity's previous tick (t - 2) was host-synchronised on its event in _collect: all five are free
            cnt.copy_(cnt_h, non_blocking=True)
            self.side.wait_stream(torch.cuda.current_stream(self.dev))        # the count upload is ordered before the collective
            with torch.cuda.stream(self.side):
                tdist.all_gather_into_tensor(counts, cnt, group=self.group)
                host.copy_(counts, non_blocking=True)
                ev.record(self.side)
            self._pending = (host, ev, tokens)
        else:
            cnt = torch.tensor([n], dtype=torch.int32)
            host = torch.empty(self.world, dtype=torch.int32)
            work = tdist.all_gather_into_tensor(host, cnt, group=self.group, async_op=True)
            self._pending = (host, work, tokens)

    def _collect(self) -> Optional[List[torch.Tensor]]:
        if self._pending is None:
            return None
        host, ready, tokens = self._pending
        self._pending = None
        import time
        t0 = time.perf_counter()
        ready.synchronize() if self.cuda else ready.wait()
        self.host_wait_s += time.perf_counter() - t0
        counts = host.tolist()
        if max(counts) == 0:
            return None
        self.payload_collectives += 1
        if tokens is not None and tokens.dtype != self.dtype:
            tokens = tokens.to(self.dtype)
        if tokens is None and self.dtype != torch.float32:
            tokens = torch.empty(0, self.d_model, dtype=self.dtype, device=self.dev)
        return _payload_allgather(tokens, counts, self.d_model, self.dev, self.group)

    def tick(self, tokens: Optional[torch.Tensor]) -> Optional[List[torch.Tensor]]:
        prev = self._collect()
        self._post(tokens)
        self.ticks += 1
        return prev

    def flush(self) -> Optional[List[torch.Tensor]]:
        return self._collect()


# ------------------------------------------------------------------------------------------------ peer-write exchange (C ABI)
class PeerWriteExchange:
    """The same tick() / flush() contract as `GatedTokenExchange`, on the library's own exchange (include/streammind_hip.h
    `sm_comm_*`, csrc/comm.hip): every rank writes its fired rows straight into a hipIpc-mapped mailbox in every peer's HBM over
    xGMI -- one hop, all links at once, no ring -- and a silent tick moves one 16-byte header per peer: no collective, no host
    allocation, no host wait (the counts of tick t are read from a pinned mirror while tick t+1 is already posted).

    `torch.distributed` is used ONCE, to move the 64-byte mailbox handles (any backend: gloo or nccl); after that the data path
    is the C ABI only, so a Level-2 (ctypes) integrator gets the same path without torch.  max_rows bounds the rows one rank
    can contribute per tick (the connector emits one token per frame: rows = frames of a tick whose gate fired)."""

    def __init__(self, d_model: int, max_rows: int = 64, group=None, dtype: torch.dtype = torch.float32,
                 device: Optional[torch.device] = None, timeout_s: float = 1800.0):
        import ctypes as C
        from . import _lib
        self._C, self.lib = C, _lib.load()
        self.group, self.d_model, self.dtype, self.max_rows = group, d_model, dtype, int(max_rows)
        # how long a tick waits for its slowest peer before it is an error.  The GPU-side collect gives up after SM_COMM_TIMEOUT_MS
        # (5 s: a spinning kernel must not pin a stream for minutes) -- that is "not yet", not an error: ranks fire on different ticks and
        # a rank decoding a 1024-token reply lags its peers by more than that, exactly where the reference's allgather simply waits
        # (process-group timeout: 30 min).  `_collect` re-issues the collect until this deadline; `late_retries` counts the re-issues.
        self.timeout_s, self.late_retries = float(timeout_s), 0
        self.world, self.rank = tdist.get_world_size(group), tdist.get_rank(group)
        self.dev = device or torch.device("cuda", torch.cuda.current_device())
        self.row_bytes = d_model * torch.empty(0, dtype=dtype).element_size()
        if self.row_bytes % 16:
            raise ValueError(f"row of {self.row_bytes} bytes: the exchange moves 16-byte words")
        # Collective construction: a rank whose local step fails still takes part in the handle exchange (with a "failed" flag), so the
        # others raise with it instead of waiting in a collective for a peer that already gave up.
        comm_dev = self.dev if tdist.get_backend(group) == "nccl" else torch.device("cpu")
        hb = self.lib.sm_comm_handle_bytes()
        self.h, err = None, None
        mine = bytes(hb)
        try:
            with torch.cuda.device(self.dev):
                h = C.c_void_p()
                _lib.check(self.lib.sm_comm_init(self.rank, self.world, self.max_rows, self.row_bytes, C.byref(h)), "sm_comm_init")
                self.h = h
                buf = (C.c_ubyte * hb)()
                _lib.check(self.lib.sm_comm_export(self.h, buf), "sm_comm_export")
                mine = bytes(buf)
        except Exception as e:      # noqa: BLE001
            err = e
        mine_t = torch.tensor([0 if err else 1] + list(mine), dtype=torch.uint8, device=comm_dev)
        every = torch.empty(self.world * (hb + 1), dtype=torch.uint8, device=comm_dev)
        tdist.all_gather_into_tensor(every, mine_t, group=group)
        every = every.cpu().reshape(self.world, hb + 1)
        if not bool(every[:, 0].all()):
            self.close()
            raise RuntimeError(f"PeerWriteExchange: mailbox creation failed on rank(s) {[r for r in range(self.world) if not every[r, 0]]}" + (f": {err!r}" if err else ""))
        try:
            with torch.cuda.device(self.dev):
                buf = (C.c_ubyte * (self.world * hb)).from_buffer_copy(bytes(every[:, 1:].reshape(-1).tolist()))
                _lib.check(self.lib.sm_comm_connect(self.h, buf), "sm_comm_connect")
        except Exception as e:      # noqa: BLE001
            err = e
        okt = torch.tensor([0 if err else 1], dtype=torch.int32, device=comm_dev)
        tdist.all_reduce(okt, op=tdist.ReduceOp.MIN, group=group)       # also the barrier: every mailbox is mapped everywhere before the first post
        if int(okt.item()) == 0:
            self.close()
            raise RuntimeError("PeerWriteExchange: a rank could not map its peers' mailboxes" + (f": {err!r}" if err else ""))
        # two result sets (tick parity): the payload of tick t is read by the caller while tick t+1 is collected
        self._payload = [torch.empty(self.world, self.max_rows, d_model, dtype=dtype, device=self.dev) for _ in range(2)]
        self._counts = (C.c_int32 * self.world)()
        self._ev = [torch.cuda.Event() for _ in range(2)]
        self.side = torch.cuda.Stream(self.dev)              # the collect WAITS for the slowest peer: it must not sit in the compute stream
        self._pending = None                                # parity of the tick posted + collected on the GPU, not yet read on the host
        self.ticks = self.payload_collectives = 0
        self.host_wait_s = 0.0

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _post(self, tokens: Optional[torch.Tensor]):
        n = 0 if tokens is None else int(tokens.shape[0])
        if n > self.max_rows:
            raise ValueError(f"{n} rows in one tick, the mailbox was sized for {self.max_rows}")
        if n:
            tokens = tokens.to(device=self.dev, dtype=self.dtype).contiguous()
        par = self.lib.sm_comm_tick(self.h) & 1              # the library's own tick counter: `ticks` below is a statistic callers may reset
        from . import _lib
        cur = torch.cuda.current_stream(self.dev)
        _lib.check(self.lib.sm_comm_post(self.h, tokens.data_ptr() if n else None, n, cur.cuda_stream), "sm_comm_post")
        self.side.wait_stream(cur)                           # post(t) before collect(t); collect(t-1) was host-synchronised in _collect
        _lib.check(self.lib.sm_comm_collect(self.h, None, self._payload[par].data_ptr(), self.side.cuda_stream), "sm_comm_collect")
        self._ev[par].record(self.side)
        self._keep = tokens                                  # the rows stay alive until the post kernel has read them
        self._pending = par

    def _collect(self) -> Optional[List[torch.Tensor]]:
        if self._pending is None:
            return None
        par, self._pending = self._pending, None
        import time
        from . import _lib
        t0 = time.perf_counter()
        missing = self._C.c_int32(-1)
        while True:
            self._ev[par].synchronize()
            rc = _lib.check(self.lib.sm_comm_poll_counts(self.h, par, self._counts, self._C.byref(missing)), "sm_comm_poll_counts")
            if rc == 0:
                break
            # a peer had not posted when the GPU-side wait gave up: the tick is NOT consumed -- collect it again (its payload arrives late,
            # never lost: the peer cannot reuse the slot before this rank has posted its next tick)
            if time.perf_counter() - t0 > self.timeout_s:
                raise RuntimeError(f"PeerWriteExchange: rank {missing.value} did not post tick {self.lib.sm_comm_tick(self.h) - 1} within {self.timeout_s:.0f} s "
                                   f"(rank {self.rank} re-issued the collect {self.late_retries} times)")
            self.late_retries += 1
            _lib.check(self.lib.sm_comm_recollect(self.h, None, self._payload[par].data_ptr(), self.side.cuda_stream), "sm_comm_recollect")
            self._ev[par].record(self.side)
        self.host_wait_s += time.perf_counter() - t0
        counts = list(self._counts)
        if max(counts) == 0:
            return None
        self.payload_collectives += 1
        return [self._payload[par][r, :counts[r]] for r in range(self.world)]

    def tick(self, tokens: Optional[torch.Tensor]) -> Optional[List[torch.Tensor]]:
        prev = self._collect()
        self._post(tokens)
        self.ticks += 1
        return prev

    def flush(self) -> Optional[List[torch.Tensor]]:
        return self._collect()

    def close(self):
        if getattr(self, "h", None):
            torch.cuda.synchronize(self.dev)
            self.lib.sm_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
