"""BASELINE configs[3] readiness on a 1-GPU box: bench.py's N > 1 path with the REAL native streams -- two ranks launched with the
driver's own command line (torch.distributed.run, 127.0.0.1 rendezvous), both on cuda:0 (SM_BENCH_ONE_DEVICE) over gloo (two RCCL
ranks cannot share a device).  Checks the one JSON line: whole-job aggregate, per-rank rates, the gated-token exchange with real frame
tokens (bf16 payload, rank-specific fire steps), the blocking all-gather of the end-to-end leg, the summed decode rate."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_one_device_gloo():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SM_BENCH_ONE_DEVICE="1", SM_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--no-aux", "--no-fp8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "plumbing_only" not in d
    assert len(d["per_rank_frames_per_s"]) == 2 and all(v > 0 for v in d["per_rank_frames_per_s"])
    assert abs(d["value"] - d["frames_per_s_per_gpu"] * 2) < 1e-2 * d["value"]
    c = d["config"]
    frames_per_step = c["frames_per_step"]
    ex = d["gated_token_exchange"]
    # steps 0..11 (warm-up included): rank 0 fires on steps 0 and 9, rank 1 on 1 and 10 -- four payload collectives, every frame token of
    # the segments since each rank's previous fire arrives
    assert ex["ticks"] == 12 and ex["payload_collectives"] == 4
    assert ex["rows_received"] == frames_per_step * (1 + 9 + 2 + 9)
    assert d["end_to_end"]["allgather_gated_tokens"] == {"calls": 2, "rows": 2 * 4 * c["frames_per_call"]}
    assert d["decode"]["tokens_per_s_all_gpus"] > d["decode"]["tokens_per_s"] > 0


@pytest.mark.parametrize("exchange", ["peer", "rccl"])
def test_bench_one_rank_forced_dist_nccl(exchange):
    """the code of BASELINE configs[3] that no 1-GPU box otherwise executes -- `init_process_group("nccl")` (RCCL), the
    collectives of the timed loop on a real device communicator, the peer-write exchange's self-test (`sm_comm_*`: hipIpc mailbox exported and
    mapped by the one rank there is) and, with SM_BENCH_EXCHANGE=rccl, the torch.distributed all-gather form -- run with ONE real rank through
    bench.py's own hook (SM_BENCH_FORCE_DIST=1, bench.py main()).  A box whose RCCL or IPC path is broken fails here, not on the first 8-GPU run."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SM_BENCH_FORCE_DIST="1", SM_BENCH_EXCHANGE=exchange, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # (the "peer" case leaves --batch at its default: the in-run schedule pick and its broadcast of rank 0's choice run on the nccl communicator too)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "9", "--warmup", "1"] + (["--batch", "28"] if exchange == "rccl" else []) + [
           "--stream-frames", "280", "--no-aux", "--no-fp8", "--no-cpu-baseline", "--no-decode", "--no-prof"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["per_rank_frames_per_s"] and len(d["per_rank_frames_per_s"]) == 1
    ex = d["gated_token_exchange"]
    fps_ = d["config"]["frames_per_step"]
    # steps 0..9 (warm-up included): the one rank fires on steps 0 and 9 -> two payload ticks carrying every frame token since the previous fire
    assert ex["ticks"] == 10 and ex["payload_collectives"] == 2 and ex["rows_received"] == fps_ * (1 + 9), ex
    if exchange == "rccl":
        assert ex["implementation"].startswith("torch.distributed all-gather (nccl)") and ex["fallback_reason"] is None, ex
    else:
        # the library's own exchange must prove itself on one rank (self-test tick), else the reason is in the line and the RCCL form took over
        assert ex["implementation"].startswith("peer_write") or (ex["fallback_reason"] and ex["implementation"].startswith("torch.distributed")), ex
        assert ex["ranks"][0]["peer_write_self_test"] is not None
    assert ex["ranks"][0]["device"] and ex["ranks"][0]["exchange"] == ex["implementation"]
    if exchange == "peer":
        assert d["config"]["schedule_pick"]["chosen_frames_per_call"] == d["config"]["frames_per_call"] in (28, 56)
