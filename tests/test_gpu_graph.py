"""BASELINE configs[4] names a "hipGraph-captured per-frame gate step".  The product path stays eager (profiles/r02_graph_ab.json,
re-measured on this round's kernels in profiles/r05_graph_ab.json by tools/graph_ab.py: replay buys nothing, the step is GPU-bound), but the C ABI must be CAPTURE-SAFE so that a deployment can capture it: no allocation,
no synchronisation, no host read-back inside a hot call once its HIP stream is warm.  These tests capture the gate step and the
decode step with hipStreamBeginCapture (torch.cuda.CUDAGraph on the stream the library launches on), replay them, and demand
results BIT-IDENTICAL to the eager calls."""
import pytest
import torch

from oracle import streammind_oracle as O
from tests.util_models import build_native, conn_gate_weights

pytestmark = pytest.mark.gpu

TV = O.VitCfg(image_size=56, patch=14, hidden=128, heads=2, mlp=256, layers=4)
TC = O.ConnCfg(mm_hidden=128, d_model=256)
TG = O.LmCfg.gate(hidden=256, heads=2, kv_heads=1, mlp=512)
TL = O.LmCfg(hidden=256, layers=2, heads=2, kv_heads=1, mlp=512, vocab=384, eps=1e-5, rope_theta=1e6)


@pytest.fixture(scope="module")
def tiny():
    Wv, Wc, Wl = O.make_vit_weights(TV, 41), conn_gate_weights(TC, TG, 86), O.make_lm_weights(TL, 44)
    return build_native(TV, TC, TG, Wv, Wc, TL, Wl, max_frames_per_call=8)


def _capture(fn):
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()                                     # warm-up ON the capture stream: the library sizes its per-HIP-stream workspaces here
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            fn()
    torch.cuda.synchronize()
    return g, side


@pytest.mark.parametrize("B", [1, 5])
def test_gate_step_captured_and_replayed_is_bit_identical(tiny, B):
    """sm_stream_reset + sm_stream_push_frames(B frames) -- tower, connector (recurrent step on the stream's state), gate, decision --
    captured once, replayed three times on fresh frame contents written into the SAME device buffer (what a ring-buffer slot is):
    logits, decisions, frame tokens and the recurrent state equal the eager call's, bit for bit."""
    m = tiny
    lib = m.lib
    frames = O.synthetic_frames(3 * B, TV.image_size, seed=5, scene_len=2).cuda()
    slot = frames[:B].clone()
    st = m.open_stream(max_frames=16, max_seq=64)
    lg = torch.empty(B, 2, device="cuda")
    dc = torch.empty(B, dtype=torch.int32, device="cuda")

    def step():
        cs = torch.cuda.current_stream().cuda_stream
        assert lib.sm_stream_reset(st.h, cs) == 0
        assert lib.sm_stream_push_frames(st.h, slot.data_ptr(), B, lg.data_ptr(), dc.data_ptr(), cs) == 0

    g, side = _capture(step)
    ref = m.open_stream(max_frames=16, max_seq=64)
    for r in range(3):
        slot.copy_(frames[r * B:(r + 1) * B])
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            g.replay()
        torch.cuda.synchronize()
        ref.reset()
        want_lg, want_dc = ref.push_frames(frames[r * B:(r + 1) * B].contiguous())
        assert torch.equal(lg, want_lg) and torch.equal(dc, want_dc), r
        tok = torch.empty(B, TC.d_model, device="cuda")
        assert lib.sm_stream_read_tokens(st.h, 0, B, tok.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        assert torch.equal(tok, ref.tokens(0, B))
        for a, b in zip(st.state(), ref.state()):
            assert torch.equal(a, b)


def test_decode_step_captured_and_replayed_is_bit_identical(tiny):
    """one greedy decode step (sm_llm_decode, n_steps = 1: fused RMSNorm + q/k/v + RoPE + KV append, decode attention, MLP, head,
    arg-max, feed-back) captured at a fixed cache position and replayed: the emitted id and the pending logits equal the eager
    step's.  (Capturing runs the host side of the call -- the position counter advances -- but no kernel; the replay then executes
    the step at the captured position.  A production graph path would keep the position on the device; capture-safety of the launch
    sequence is what is shown here.)"""
    m = tiny
    lib = m.lib
    g0 = torch.Generator().manual_seed(3)
    ids = torch.randint(3, TL.vocab, (40,), generator=g0).to(torch.int32).cuda()
    eager, cap = m.open_stream(max_frames=8, max_seq=128), m.open_stream(max_frames=8, max_seq=128)
    for s in (eager, cap):
        s.prefill(ids)
        s.decode(3)
    kv0 = eager.kv_len
    want_id = int(eager.decode(1)[0])
    want_lg = eager.logits()[0].clone()
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    warm = m.open_stream(max_frames=8, max_seq=128)          # sizes the per-HIP-stream workspaces of `side` before the capture
    with torch.cuda.stream(side):
        warm.prefill(ids)
        warm.decode(2)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gph, stream=side):
            assert lib.sm_llm_decode(cap.h, 1, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert cap.kv_len == kv0 + 1 and int(out[0]) == 0        # host counter moved, nothing ran
    with torch.cuda.stream(side):
        gph.replay()
    torch.cuda.synchronize()
    assert int(out[0]) == want_id
    assert torch.equal(cap.logits()[0], want_lg)


@pytest.mark.parametrize("B", [1, 16])
def test_config4_full_size_fp8_weights_captured_step_is_bit_identical(B):
    """BASELINE configs[4] as ONE thing (round 5): the FULL-SIZE model (CLIP-ViT-L/14-336 tower, 872 M-parameter gate and Mistral-7B, gate + LLM weights in
    fp8 e4m3 with per-row scales: weights_fp8 = 2) with the per-frame gate step -- sm_stream_reset + sm_stream_push_frames(B frames) -- captured by
    hipStreamBeginCapture and replayed on fresh frames in the same ring slot: logits, decisions, frame tokens and the recurrent state BIT-IDENTICAL to
    the eager call; then one fp8-weight Mistral-7B decode step captured at a fixed cache position and replayed: id and pending logits equal the eager
    step's.  Random weights of the true shapes (the bench's generators: no oracle needed for an identity); the timing A/B of the same captures is
    tools/graph_ab.py (profiles/r05_graph_ab.json, r05_graph_ab_fp8.json)."""
    import bench
    from streammind_amd.native import NativeModel, PathConfig
    cfg = PathConfig(llm_layers=32, max_frames_per_call=16, weights_fp8=2)
    m = NativeModel(cfg)
    bench.random_weights_into(m, cfg, 1)
    bench.random_llm_weights_into(m, cfg, 2)
    m.finalize()
    lib = m.lib
    frames = bench.synthetic_frames_gpu(3 * B, 336, 1, 0)
    slot = frames[:B].clone()
    st = m.open_stream(max_frames=32, max_seq=512)
    lg = torch.empty(B, 2, device="cuda")
    dc = torch.empty(B, dtype=torch.int32, device="cuda")

    def step():
        cs = torch.cuda.current_stream().cuda_stream
        assert lib.sm_stream_reset(st.h, cs) == 0
        assert lib.sm_stream_push_frames(st.h, slot.data_ptr(), B, lg.data_ptr(), dc.data_ptr(), cs) == 0

    g, side = _capture(step)
    ref = m.open_stream(max_frames=32, max_seq=512)
    for r in range(3):
        slot.copy_(frames[r * B:(r + 1) * B])
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            g.replay()
        torch.cuda.synchronize()
        ref.reset()
        want_lg, want_dc = ref.push_frames(frames[r * B:(r + 1) * B].contiguous())
        assert torch.isfinite(want_lg).all()
        ref.reset()
        again_lg, _ = ref.push_frames(frames[r * B:(r + 1) * B].contiguous())
        assert torch.equal(again_lg, want_lg), ("eager call not repeatable", r, again_lg, want_lg)
        assert torch.equal(lg, want_lg) and torch.equal(dc, want_dc), (r, lg, want_lg)
        tok = torch.empty(B, cfg.conn_d_model, device="cuda")
        assert lib.sm_stream_read_tokens(st.h, 0, B, tok.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        assert torch.equal(tok, ref.tokens(0, B))
        for a, b in zip(st.state(), ref.state()):
            assert torch.equal(a, b)
    if B == 1:
        gen = torch.Generator(device="cuda").manual_seed(3)
        ids = torch.randint(3, cfg.llm_vocab, (200,), generator=gen, device="cuda", dtype=torch.int32)
        eager, cap, warm = (m.open_stream(max_frames=8, max_seq=512) for _ in range(3))
        for s_ in (eager, cap):
            s_.prefill(ids)
            s_.decode(3)
        kv0 = eager.kv_len
        want_id = int(eager.decode(1)[0])
        want_lg = eager.logits()[0].clone()
        out = torch.zeros(1, dtype=torch.int32, device="cuda")
        side2 = torch.cuda.Stream()
        with torch.cuda.stream(side2):
            warm.prefill(ids)
            warm.decode(2)
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side2):
            with torch.cuda.graph(gph, stream=side2):
                assert lib.sm_llm_decode(cap.h, 1, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        assert cap.kv_len == kv0 + 1
        with torch.cuda.stream(side2):
            gph.replay()
        torch.cuda.synchronize()
        assert int(out[0]) == want_id and torch.equal(cap.logits()[0], want_lg)
    m.close()
