"""Throughput-mode stream runtime: the caller loop of eval/video_score_stream_demo.py:258-302 re-organised around
O(1)-per-frame device state instead of one blocking call per frame.

    decoded u8 frames (host) --pinned ring--> async H2D on a copy stream --> device ring
        --> sm_stream_push_frames (ViT batch + connector scan + gate, one launch sequence per batch)
        --> ONE device->host read of the batch's decisions
        --> for every fired frame, in order: splice + prefill (KV prefix reuse) + greedy decode, prompt growth

Results are identical to the frame-at-a-time loop (`streammind_amd.stream_infer` per frame): the gate of frame t depends only on
frames <= t (the Mamba scan is causal and the gate sees one token), never on what the LLM said, so perceiving a batch
ahead of the replies changes nothing but latency.  The reply for a fire at frame t is generated from exactly the
tokens [0, t] and the prompt grown by the earlier replies, as in the reference.

`run(..., overlap_replies=True)` additionally moves the replies onto an LLM LANE -- a second HIP stream: splice + prefill + greedy
decode of a reply are enqueued there in chunks while the perception stream keeps consuming frames.  Decode at batch 1 is
HBM-bound (~60 % of the HBM peak, ~4 % of the matrix pipes), the tower the opposite (~40 % of the MFMA peak, a fifth of the HBM
bandwidth): sharing the chip they take each other's idle resource, and -- what matters for a live stream -- the gate keeps
deciding on new frames while a reply is being written (the reference's loop, eval/video_score_stream_demo.py:283-299, perceives
nothing for the ~0.8 s a 256-token reply takes).  A fire that arrives while a reply is still being decoded waits for it (its
prompt contains that reply): replies, gate logits and the grown prompt are bit-identical to the serial order."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterable, Iterator, List, Optional, Tuple

import torch

from .constants import MMODAL_TOKEN_INDEX
from .mm_utils import KeywordsStoppingCriteria, tokenizer_MMODAL_token


class FrameRing:
    """Pinned-host ring of u8 HWC frames + a device ring; `push` stages a batch with an async copy on a dedicated HIP
    stream and returns the device view plus the event the compute stream must wait on."""

    def __init__(self, slots: int, frames_per_slot: int, height: int, width: int, device: torch.device):
        self.slots, self.fps = slots, frames_per_slot
        self.host = torch.empty(slots, frames_per_slot, height, width, 3, dtype=torch.uint8).pin_memory()
        self.dev = torch.empty(slots, frames_per_slot, height, width, 3, dtype=torch.uint8, device=device)
        self.copy_stream = torch.cuda.Stream(device=device)
        self.ready = [torch.cuda.Event() for _ in range(slots)]
        self.free = [torch.cuda.Event() for _ in range(slots)]
        self.head = 0

    def push(self, frames: torch.Tensor) -> Tuple[torch.Tensor, torch.cuda.Event, int]:
        n = frames.shape[0]
        assert n <= self.fps and frames.dtype == torch.uint8
        s = self.head
        self.head = (self.head + 1) % self.slots
        self.free[s].synchronize()                      # the slot's previous consumer finished (no-op the first time round)
        if frames.is_pinned():                          # a decoder that writes into page-locked memory: straight to the device
            src = frames
        else:                                           # pageable source: staged through the slot's pinned buffer (a host memcpy)
            self.host[s, :n].copy_(frames)
            src = self.host[s, :n]
        with torch.cuda.stream(self.copy_stream):
            self.dev[s, :n].copy_(src, non_blocking=True)
            self.ready[s].record(self.copy_stream)
        return self.dev[s, :n], self.ready[s], s

    def release(self, slot: int) -> None:
        self.free[slot].record(torch.cuda.current_stream())


@dataclass
class StreamEvent:
    frame_index: int            # 1-based count of frames seen when the gate fired (== interval id)
    text: str
    new_ids: List[int]


@dataclass
class StreamStats:
    frames: int = 0
    fires: int = 0
    gate_logits: List[torch.Tensor] = field(default_factory=list)


class StreamingSession:
    """One video stream on one GPU.  `model` is a streammind_amd.model.Videollama2MistralForCausalLM."""

    def __init__(self, model, tokenizer, batch_frames: int = 8, max_new_tokens: int = 1024, ring_slots: int = 3,
                 keep_logits: bool = False, source_hw: Optional[Tuple[int, int]] = None, aspect_ratio: Optional[str] = "pad"):
        """source_hw: (H, W) of the decoded frames when they are not vit_image x vit_image -- they are then staged at their
        native size and resized on the GPU by the ingest front-end (expand2square when aspect_ratio == "pad", PIL-exact
        bicubic, centre crop: mm_utils.py:452-464) before perception, i.e. exactly what `process_video` would have produced."""
        self.model, self.tok = model, tokenizer
        cfg = model.native.cfg
        self.batch = min(batch_frames, cfg.max_frames_per_call)
        self.max_new = max_new_tokens
        self.image = cfg.vit_image
        self.src_hw = tuple(source_hw) if source_hw is not None else (cfg.vit_image, cfg.vit_image)
        self.pad_square = aspect_ratio == "pad"
        self.pad_rgb = tuple(int(x * 255) for x in cfg.img_mean)
        self.ring = FrameRing(ring_slots, self.batch, self.src_hw[0], self.src_hw[1], model.device)
        self.prompt: Optional[str] = None
        self.stats = StreamStats()
        self.keep_logits = keep_logits

    def _initial_prompt(self) -> str:
        from .conversation import conv_templates
        conv = conv_templates["mistral_instruct"].copy()
        conv.append_message(conv.roles[0], "<video>\n")
        conv.append_message(conv.roles[1], None)
        return conv.get_prompt()

    def _reply_inputs(self, upto_frame: int):
        m = self.model
        m.interval_id_list.append(upto_frame)
        input_ids = tokenizer_MMODAL_token(self.prompt, self.tok, MMODAL_TOKEN_INDEX["VIDEO"], return_tensors="pt").unsqueeze(0)
        crit = KeywordsStoppingCriteria(["</s>"], self.tok, input_ids)
        return m._expand(input_ids[0].tolist()), crit

    def _reply_done(self, upto_frame: int, new_ids: List[int]) -> StreamEvent:
        text = self.tok.batch_decode([new_ids], skip_special_tokens=True)[0].strip()
        self.prompt += " " + text + " </s>[INST] <video>\n [/INST]"          # video_score_stream_demo.py:124
        return StreamEvent(upto_frame, text, new_ids)

    def _reply(self, upto_frame: int) -> StreamEvent:
        seq, crit = self._reply_inputs(upto_frame)
        return self._reply_done(upto_frame, self.model._generate(seq, self.max_new, [crit]))

    # ---- replies on the LLM lane (run(overlap_replies=True)).  One reply is in flight at a time; its decode chunks are enqueued
    # `lane_depth` ahead on the lane's HIP stream and their ids come back through pinned memory behind an event, so the host never
    # blocks on the lane while frames are waiting (it does when the frames have run out: _pump(block=True)).
    def _lane_start(self, upto_frame: int) -> None:
        m = self.model
        seq, crit = self._reply_inputs(upto_frame)
        self._llm.wait_stream(torch.cuda.current_stream())    # the lane starts behind everything the perception stream has been given
        with torch.cuda.stream(self._llm):                    # (sm_llm_prefill also orders it behind the newest side-stream pass)
            budget = m._begin_generate(seq, self.max_new)     # prefill behind the cached prefix
        self._active = {"frame": upto_frame, "seq_len": len(seq), "crit": crit, "budget": budget, "out": [], "issued": 0, "chunks": []}

    def _lane_issue(self) -> None:
        a, m = self._active, self.model
        while len(a["chunks"]) < self.lane_depth and a["issued"] < a["budget"]:
            n = min(m.decode_chunk, a["budget"] - a["issued"])
            # pinned landing buffers and events are allocated ONCE per session (lane_depth + 1 of each: a pinned allocation is a
            # synchronous hipHostMalloc, this is the hot overlap path) and reused round-robin; a slot is free again once its chunk was taken
            slot = self._lane_slots[self._lane_next % len(self._lane_slots)]
            self._lane_next += 1
            with torch.cuda.stream(self._llm):
                ids = m.stream.decode(n)
                host = slot[0][:n]
                host.copy_(ids, non_blocking=True)
                slot[1].record(self._llm)
            a["issued"] += n
            a["chunks"].append((host, slot[1], ids))

    def _pump(self, block: bool) -> Iterator[StreamEvent]:
        """advance the lane: start the next queued fire, keep `lane_depth` decode chunks enqueued, take the chunks that have
        finished (every chunk when `block`), close the reply at its stop.  Everything of the LLM is on ONE HIP stream, in order:
        chunks enqueued past a stop are speculative -- their ids are dropped, the cache length is cut back by _accept_tokens (as
        for the speculative tail of a serial chunk) and the next prefill simply follows them on the lane."""
        m = self.model
        while True:
            if self._active is None:
                if not self._fire_q:
                    return
                self._lane_start(self._fire_q.pop(0))
            a = self._active
            self._lane_issue()
            closed = not a["chunks"]                           # nothing (left) to decode: the context filled the cache
            if not closed:
                host, ev, _ = a["chunks"][0]
                if not block and not ev.query():
                    return                                     # the lane is busy; perception goes on
                ev.synchronize()
                a["chunks"].pop(0)
                _, done = m._accept_tokens(a["out"], host.tolist(), a["seq_len"], [a["crit"]])
                closed = done or len(a["out"]) >= a["budget"]
            if closed:
                self._active = None
                yield self._reply_done(a["frame"], a["out"])

    def _issue(self, frames: torch.Tensor, pipelined: bool):
        """stage one batch and enqueue its perception; returns what `_collect` needs (no host sync here)"""
        if self.prompt is None:
            self.prompt = self._initial_prompt()
        dev_frames, ready, slot = self.ring.push(frames)
        torch.cuda.current_stream().wait_event(ready)
        base = self.model.stream.num_frames
        if self.src_hw != (self.image, self.image):
            from . import native
            dev_frames = native.ingest_frames(dev_frames.contiguous(), self.pad_square, self.image, self.pad_rgb)
        push = self.model.stream.push_frames_pipelined if pipelined else self.model.stream.push_frames
        logits, dec = push(dev_frames)
        self.ring.release(slot)
        return logits, dec, base, (self.model.stream.last_ticket if pipelined else None)

    def _collect(self, handle, fires_only: bool = False):
        """the one host sync of a batch (its decisions), then the replies its frames fired, in order (fires_only: the fired frame
        numbers instead -- the caller runs the replies on the LLM lane)"""
        logits, dec, base, ticket = handle
        if ticket is not None:
            # read back on a stream of its own, ordered behind THIS batch's pass only: neither the host nor the compute stream
            # waits for the tower / pass of the batch issued ahead (sm_stream_join would wait for the newest pass)
            if getattr(self, "_rd", None) is None:
                self._rd = torch.cuda.Stream(self.model.device)
            with torch.cuda.stream(self._rd):
                self.model.stream.join(ticket)
                dec.record_stream(self._rd); logits.record_stream(self._rd)
                dec_h = dec.to("cpu", non_blocking=False)
                lg_h = logits.to("cpu", non_blocking=False) if self.keep_logits else None
            dec_host = dec_h.tolist()
        else:
            dec_host = dec.cpu().tolist()
            lg_h = logits.cpu() if self.keep_logits else None
        if self.keep_logits:
            self.stats.gate_logits.append(lg_h)
        self.stats.frames += len(dec_host)
        events = []
        for j, d in enumerate(dec_host):
            if d == 1:
                self.stats.fires += 1
                events.append(base + j + 1 if fires_only else self._reply(base + j + 1))
        return events

    def feed(self, frames: torch.Tensor) -> List[StreamEvent]:
        """frames: u8 [n,H,W,3] on the HOST (n <= batch_frames).  Returns the replies fired by these frames, in order."""
        return self._collect(self._issue(frames, pipelined=False))

    def run(self, frames: Iterable[torch.Tensor], overlap_replies: bool = False, lane_depth: int = 2) -> Iterator[StreamEvent]:
        """frames: iterable of u8 [H,W,3] host tensors (the decoded stream).  Yields replies as they fire.

        One batch of look-ahead: batch i+1 is staged and its tower enqueued BEFORE the decisions of batch i are read back, and
        the connector + gate pass of every batch runs on the stream's side HIP stream (sm_stream_push_frames_pipelined); the decision
        read of batch i waits for batch i's pass only (per-call ticket, read-back stream), so batch i+1 keeps the GPU busy while the
        host looks at batch i, and the memory-bound pass overlaps the next tower.  A reply for a fire in batch i
        is generated from the tokens [0, t] of its own frame: frames perceived ahead change nothing but latency.

        overlap_replies: replies are decoded on the LLM lane while perception goes on (module docstring); same events, same order."""
        buf: List[torch.Tensor] = []
        pending = None
        self._fire_q, self._active, self.lane_depth = [], None, max(1, lane_depth)
        if overlap_replies and getattr(self, "_llm", None) is None:
            self._llm = torch.cuda.Stream(self.model.device)
        if overlap_replies and len(getattr(self, "_lane_slots", ())) != self.lane_depth + 1:
            self._lane_slots = [(torch.empty(max(1, self.model.decode_chunk), dtype=torch.int32).pin_memory(), torch.cuda.Event())
                                for _ in range(self.lane_depth + 1)]
        self._lane_next = 0

        def batches():
            nonlocal buf
            for f in frames:
                buf.append(f)
                if len(buf) == self.batch:
                    yield torch.stack(buf)
                    buf = []
            if buf:
                yield torch.stack(buf)

        def collect(handle):
            if not overlap_replies:
                yield from self._collect(handle)
                return
            self._fire_q.extend(self._collect(handle, fires_only=True))
            yield from self._pump(block=False)
        try:
            for b in batches():
                handle = self._issue(b, pipelined=True)
                if pending is not None:
                    yield from collect(pending)
                elif overlap_replies:
                    yield from self._pump(block=False)
                pending = handle
            if pending is not None:
                yield from collect(pending)
            if overlap_replies:
                yield from self._pump(block=True)             # the stream has ended: finish what the lane still owes
        finally:
            # also when the consumer abandons the generator early: whatever is still in flight on the lane is ordered in front of the
            # next thing the compute stream does (a later serial feed() / _reply must not race it on the KV cache), the lane's queue is dropped
            if overlap_replies and getattr(self, "_llm", None) is not None:
                torch.cuda.current_stream().wait_stream(self._llm)
            self._fire_q, self._active = [], None


class MultiStreamSession:
    """S concurrent video streams on one GPU, one frame per stream per tick -- the reference's per-frame loop
    (eval/video_score_stream_demo.py:283-299) for MANY streams at once.

        tick(frames[S]) : ONE ViT batch + one connector / gate weight pass for all streams (sm_group_push_frames), one device->host
                          read of the S decisions; the streams whose gate fired get their replies TOGETHER: each prefills its own
                          grown prompt (KV prefix reuse), then all of them decode in lock step through sm_group_llm_decode -- one
                          pass over the LLM weights per step -- until each hits its own stop.

    Per stream the results are those of its own `streammind_amd.stream_infer` loop (same gate decisions, same prompt growth; the
    reply ids are the greedy ids of the same logits up to fp32 summation order).  `models`: one
    Videollama2MistralForCausalLM per stream, all built on the SAME NativeModel."""

    def __init__(self, models, tokenizer, max_new_tokens: int = 1024, decode_chunk: int = 16, continuous: bool = False):
        """continuous=False: a tick returns when the replies it started are complete (every stream waits for them: the lock-step form).
        continuous=True: replies stay IN FLIGHT across ticks -- a tick is one perception pass for every stream plus `decode_chunk` decode steps of all
        the streams that are replying at that moment, whichever tick started them (one weight pass per step for all of them: a reply of 256 tokens no
        longer stops the other streams' frames for 256 steps, and replies started ticks apart share their weight passes); a stream that fires again
        while it is still replying gets that reply right behind the running one, on the frame interval it fired at.  Perception does not depend on the
        replies (the gate reads the connector state only), so per stream the events are those of the lock-step form -- they are handed out by the tick
        in which they complete; `flush()` finishes what is in flight."""
        assert len(models) >= 1 and all(m.native is models[0].native for m in models), "the streams must share one NativeModel"
        self.models, self.tok = list(models), tokenizer
        self.native = models[0].native
        self.group = self.native.open_group([m.stream for m in models])
        self.max_new, self.chunk = max_new_tokens, decode_chunk
        self.prompts: List[Optional[str]] = [None] * len(models)
        self.stats = StreamStats()
        self.continuous = continuous
        self._replying = {}                                # stream index -> state of the reply in flight
        self._pending = [[] for _ in models]               # stream index -> frame positions of fires that wait for the running reply

    def _initial_prompt(self) -> str:
        from .conversation import conv_templates
        conv = conv_templates["mistral_instruct"].copy()
        conv.append_message(conv.roles[0], "<video>\n")
        conv.append_message(conv.roles[1], None)
        return conv.get_prompt()

    def tick(self, frames: torch.Tensor) -> List[Tuple[int, StreamEvent]]:
        """frames: u8 [S, H, W, 3] (host or device), frame t of every stream.  -> [(stream index, reply event)] of this tick."""
        S = len(self.models)
        assert frames.shape[0] == S
        logits, dec = self.group.push_frames(frames.to(self.native.device).contiguous())
        dec_host = dec[:, 0].cpu().tolist()               # the one host sync of the tick
        self.last_gate_logits = logits[:, 0]
        self.stats.frames += S
        fired = [i for i, d in enumerate(dec_host) if d == 1]
        self.stats.fires += len(fired)
        if self.continuous:
            for i in fired:
                upto = self.models[i].stream.num_frames
                if i in self._replying:
                    self._pending[i].append(upto)            # behind the running reply (its text belongs to this one's prompt)
                else:
                    self._start_reply(i, upto)
            return self._decode_round(self.chunk)
        if not fired:
            return []
        # ---- every fired stream: its own splice + prefill (prompt growth and KV prefix reuse as in the single-stream loop)
        for i in fired:
            self._start_reply(i, self.models[i].stream.num_frames)
        # ---- all of them decode together, to the end
        events = []
        while self._replying:
            events += self._decode_round(self.chunk)
        events.sort(key=lambda e: fired.index(e[0]))
        return events

    def _start_reply(self, i: int, upto: int) -> None:
        m = self.models[i]
        if self.prompts[i] is None:
            self.prompts[i] = self._initial_prompt()
        m.interval_id_list.append(upto)
        input_ids = tokenizer_MMODAL_token(self.prompts[i], self.tok, MMODAL_TOKEN_INDEX["VIDEO"], return_tensors="pt").unsqueeze(0)
        seq = m._expand(input_ids[0].tolist())
        budget = m._begin_generate(seq, self.max_new)
        self._replying[i] = {"seq_len": len(seq), "budget": budget, "out": [], "crit": [KeywordsStoppingCriteria(["</s>"], self.tok, input_ids)], "upto": upto}

    def _decode_round(self, steps: int) -> List[Tuple[int, StreamEvent]]:
        """up to `steps` decode steps of every reply in flight (one sm_group_llm_decode call: one weight pass per step for all of them); the replies
        that complete are closed (prompt growth as video_score_stream_demo.py:124) and a fire that waited behind one starts right away"""
        if not self._replying:
            return []
        S = len(self.models)
        active = sorted(self._replying)
        n = min([steps] + [self._replying[i]["budget"] - len(self._replying[i]["out"]) for i in active])
        ids = self.group.decode(n, active=[i in self._replying for i in range(S)]).cpu().tolist()
        events = []
        for i in active:
            st = self._replying[i]
            _, done = self.models[i]._accept_tokens(st["out"], ids[i], st["seq_len"], st["crit"])
            if done or len(st["out"]) >= st["budget"]:
                del self._replying[i]
                text = self.tok.batch_decode([st["out"]], skip_special_tokens=True)[0].strip()
                self.prompts[i] += " " + text + " </s>[INST] <video>\n [/INST]"          # video_score_stream_demo.py:124
                events.append((i, StreamEvent(st["upto"], text, st["out"])))
                if self._pending[i]:
                    self._start_reply(i, self._pending[i].pop(0))
        return events

    def flush(self) -> List[Tuple[int, StreamEvent]]:
        """continuous mode: decode until no reply is in flight (end of the streams) -> the events that completed"""
        events = []
        while self._replying:
            events += self._decode_round(self.chunk)
        return events
