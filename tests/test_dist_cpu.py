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
