"""N > 1 plumbing on CPU: world_size-2 gloo processes (rendezvous on 127.0.0.1)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from streammind_amd.dist import allgather_gated_tokens, partition_streams


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
