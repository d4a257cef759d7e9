"""Parity at the BASELINE configs' own sizes (VERDICT r3 "next round" #1).

  (a) configs[1] / configs[4] horizons: the recurrent connector + gate, one frame per call, for 1800 and 3600 steps at full size,
      against the oracle's full scan over the whole history (what the reference recomputes every frame);
  (b) configs[2]: CLIP-ViT-L/14-336 AND Mistral-7B in ONE sm_model -- 56 frames, two scheduled fires, two 256-token replies with
      KV prefix reuse, every decode step against the oracle;
  (c) greedy ids on weights with planted top-2 margins: all 256 ids EQUAL, bf16 and llm_fp16; flips / 256 reported for the
      random-weight case in both modes.
"""
import pytest
import torch

from oracle import streammind_oracle as O
from tests.util_models import build_native, conn_gate_weights, path_config

pytestmark = pytest.mark.gpu


def maxdiff(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


def _stream_pooled(T: int, seed: int) -> torch.Tensor:
    """pooled tower features of a T-frame stream: piecewise-constant 'scenes' (30-240 frames, the Ego4D-shape gaps of SURVEY 8d at
    30 fps) + per-frame jitter, at the magnitude real pooled CLIP features have in the full-size tests (|x| up to a few units)."""
    g = torch.Generator().manual_seed(seed)
    out = torch.empty(T, 1024)
    t = 0
    while t < T:
        n = int(torch.randint(30, 241, (1,), generator=g))
        scene = torch.randn(1024, generator=g) * 0.6
        out[t:t + n] = scene
        t += n
    return out + torch.randn(T, 1024, generator=g) * 0.05


@pytest.fixture(scope="module")
def conn_gate_full():
    ccfg, gcfg = O.ConnCfg(), O.LmCfg.gate()
    Wc = conn_gate_weights(ccfg, gcfg, 4242)
    vcfg = O.VitCfg(image_size=28, patch=14, hidden=1024, heads=16, mlp=64, layers=2)
    m = build_native(vcfg, ccfg, gcfg, O.make_vit_weights(vcfg, 1), Wc)
    return m, Wc, ccfg, gcfg


@pytest.mark.parametrize("T", [1800, 3600])
def test_recurrent_connector_and_gate_one_frame_per_call_vs_full_scan(conn_gate_full, T):
    """The central substitution at the horizon the configs name: BASELINE configs[1] is an 1800-frame stream, configs[4] 3600.  The
    reference re-runs Video_Mamba_seq over the WHOLE history for every new frame (videollama2_arch.py:186-191, builder.py:405,
    547-562); this build carries conv_state / ssm_state and does one recurrent step per frame (sm_stream_push_pooled, M = 1).
    FULL SIZE (1024 -> 4096, d_inner 8192, d_state 16; 872 M-parameter gate).  Against O.connector_scan over all T frames:
      * every one of the T frame tokens within 1e-4,
      * gate logits within 1e-3 at frames {1, 2, 600, 601, 1799, 1800[, 3600]} and every 40th frame; decisions equal outside 2e-3,
      * conv_state and ssm_state after the last frame within 1e-5 (relative to the largest entry) of the scan's final state,
      * the same stream pushed 8 frames per call is bit-identical (tokens, logits, states); 16 and 28 per call (the bench's step) agree to
        fp32 summation noise (5e-5) and meets the same bounds against the oracle."""
    m, Wc, ccfg, gcfg = conn_gate_full
    torch.set_num_threads(max(16, torch.get_num_threads()))
    pooled = _stream_pooled(T, 77)
    pg = pooled.cuda()
    s = m.open_stream(max_frames=T, max_seq=64)
    lgs, dcs = [], []
    for t in range(T):
        lg, dc = s.push_pooled(pg[t:t + 1])
        lgs.append(lg); dcs.append(dc)
    lg1, dc1 = torch.cat(lgs).cpu(), torch.cat(dcs).cpu()
    tok1 = s.tokens().cpu()
    conv1, ssm1 = (x.cpu() for x in s.state())
    tok_ref, st_ref = O.connector_scan(pooled, Wc, ccfg, return_state=True)
    dt = (tok1 - tok_ref).abs().amax(dim=1)
    named = [f for f in (1, 2, 600, 601, 1799, 1800, 3600) if f <= T]
    idx = sorted(set([f - 1 for f in named] + list(range(39, T, 40))))
    ref_lg = O.gate_logits_shortcut(tok_ref[idx], Wc, gcfg)
    dl = (lg1[idx] - ref_lg).abs().amax(dim=1)
    dconv, dssm = maxdiff(conv1, st_ref.conv), maxdiff(ssm1, st_ref.ssm)
    print(f"T={T}: tokens max|diff| {float(dt.max()):.3e} (first 100: {float(dt[:100].max()):.3e}, last 100: {float(dt[-100:].max()):.3e}; max |token| "
          f"{float(tok_ref.abs().max()):.2f}); gate logits max|diff| over {len(idx)} frames {float(dl.max()):.3e}, at frames {named}: "
          f"{[f'{float(dl[idx.index(f - 1)]):.1e}' for f in named]}; conv_state {dconv:.2e} (max |x| {float(st_ref.conv.abs().max()):.2f}) ssm_state {dssm:.2e} (max |h| {float(st_ref.ssm.abs().max()):.2f}); "
          f"fires {int(dc1.sum())}/{T}")
    assert float(dt.max()) < 1e-4
    assert float(dl.max()) < 1e-3
    for j, i in enumerate(idx):
        if abs(float(ref_lg[j, 1] - ref_lg[j, 0])) > 2e-3:
            assert int(dc1[i]) == O.gate_decision(ref_lg[j]), i
    assert dconv < 1e-5 * max(1.0, float(st_ref.conv.abs().max())) and dssm < 1e-5 * max(1.0, float(st_ref.ssm.abs().max()))
    # the same stream pushed 8 frames per call (the <= 9-row weight-streaming kernels: same accumulation order per row as one
    # frame per call) is BIT-identical; 16 and 28 frames per call (the bench's step: from 10 rows -- round 5, it was 17 -- the kernels share the
    # activations through LDS and sum K slices in another order) agree to fp32 summation noise and meet the same bounds against the oracle
    for chunk in (8, 16, 28):
        s2 = m.open_stream(max_frames=T, max_seq=64)
        lg2 = torch.cat([s2.push_pooled(pg[t:t + chunk].contiguous())[0] for t in range(0, T, chunk)]).cpu()
        tok2 = s2.tokens().cpu()
        conv2, ssm2 = (x.cpu() for x in s2.state())
        if chunk == 8:
            assert torch.equal(lg2, lg1) and torch.equal(tok2, tok1) and torch.equal(conv2, conv1) and torch.equal(ssm2, ssm1)
        else:
            d_tok, d_lg = maxdiff(tok2, tok1), maxdiff(lg2, lg1)
            print(f"  {chunk} frames per call vs 1: tokens {d_tok:.2e}, gate logits {d_lg:.2e}; vs oracle: tokens {maxdiff(tok2, tok_ref):.2e}, logits {maxdiff(lg2[idx], ref_lg):.2e}")
            assert d_tok < 5e-5 and d_lg < 5e-5
            assert maxdiff(tok2, tok_ref) < 1e-4 and maxdiff(lg2[idx], ref_lg) < 1e-3
            assert maxdiff(ssm2, st_ref.ssm) < 1e-5 * max(1.0, float(st_ref.ssm.abs().max()))


class _LazyF32(dict):
    """16-bit weights on the host, widened to fp32 one tensor at a time for the oracle"""

    def __getitem__(self, k):
        return super().__getitem__(k).float()


def _seeded_llm_into(m, lcfg, seed, Wl, embed_std=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    for name, shp in O.lm_weight_shapes(lcfg, "", True).items():
        if "layernorm" in name or name.endswith("model.norm.weight"):
            w = (1.0 + 0.1 * torch.randn(*shp, generator=g, device="cuda")).to(torch.bfloat16)
            m.load_tensor(name, w.float())
        else:
            std = embed_std if "embed_tokens" in name else shp[-1] ** -0.5
            w = (torch.randn(*shp, generator=g, device="cuda") * std).to(torch.bfloat16)
            m.load_tensor(name, w)
        Wl[name] = w.cpu()
        del w


def _decode_check(s, Wl, lcfg, emb, n_new, rel_tol, what, prec=None):
    """tests/test_gpu_path.py::_greedy_decode_check: every step's logits against the oracle evaluating the stream's own token
    sequence in one causal pass; ids equal to the oracle's arg-max outside near-ties.  Returns (ids, flips)."""
    prec = O.MIXED if prec is None else prec
    table = Wl["model.embed_tokens.weight"]
    ids, logits = [], []
    for j in range(n_new):
        lg, nt = s.logits()
        logits.append(lg.cpu())
        ids.append(int(s.decode(1)[0]))
        assert ids[-1] == int(nt)
    emb_all = torch.cat([emb, table[torch.tensor(ids[:-1])]]) if n_new > 1 else emb
    ref = O.lm_forward(emb_all, Wl, lcfg, O.KVCache(), prec, last_only=False)[emb.shape[0] - 1:]
    tol = rel_tol * max(1.0, float(ref[0].abs().max()))
    flips, worst, min_margin = 0, 0.0, float("inf")
    for j in range(n_new):
        d = maxdiff(logits[j], ref[j])
        worst = max(worst, d)
        assert d < tol, (what, j, d, tol)
        margin = float(torch.topk(ref[j], 2).values.diff().abs())
        min_margin = min(min_margin, margin)
        if int(torch.argmax(ref[j])) != ids[j]:
            flips += 1
            if flips <= 4:
                top = torch.topk(ref[j], 2)
                print(f"  flip at step {j}: oracle top-2 ids {top.indices.tolist()} logits {[round(float(v), 4) for v in top.values]} (margin {margin:.3e}); "
                      f"stream chose {ids[j]} with its own logits {[round(float(logits[j][int(i)]), 4) for i in top.indices]}")
            assert margin < 2 * tol, (what, j, ids[j], int(torch.argmax(ref[j])), margin)
    print(f"{what}: {n_new} steps, worst logit diff {worst:.3e} (tol {tol:.3e}), smallest oracle top-2 margin {min_margin:.3e}, "
          f"{flips} of {n_new} ids differ from the oracle's arg-max (all inside near-ties)")
    _decode_check.last_min_margin, _decode_check.last_tol = min_margin, tol
    return ids, flips


def test_vit_l_and_mistral_7b_in_one_model_56_frames_two_256_token_replies():
    """BASELINE configs[2] as ONE model: CLIP-ViT-L/14-336 (23 layers run) + full-size connector + 872 M-parameter gate + Mistral-7B
    (32 distinct layers, vocab 32 000) in the same sm_model.  56 frames in two 28-frame calls; a scheduled fire after each call
    (scheduled fires for throughput configs); each fire = splice of the stream's OWN frame tokens into the prompt,
    prefill, a 256-token greedy reply.  Fire 2 prefills only the new positions behind the cached prefix + reply 1 (prefix reuse;
    the reference re-prefills everything, videollama2_mistral.py:413,426-431).
    The oracle is end to end and independent: ITS tower (bf16-rounding mode) -> ITS connector scan -> ITS tokens spliced -> from-
    scratch prefill at each fire, all positions.  Checked: gate logits of all 56 frames against the bf16-mode oracle (the bench's
    precision) at 2e-3 -- measured 1.15e-3 on these frames; 8.8e-4 on the frames of test_full_size_28_frames_bf16_tower_vs_bf16_mode_
    oracle, which asserts the north-star's 1e-3: two correct bf16-operand towers differ by up to 1.5e-3 on these logits,
    the fp16 tower that load_pretrained_model selects sits 1.3e-4 from fp32 --, decisions outside twice that, frame tokens, and
    EVERY one of the 2 x 256 decode steps (logits 3e-2 x scale, ids equal outside near-ties)."""
    vcfg, ccfg, gcfg = O.VitCfg(), O.ConnCfg(), O.LmCfg.gate()
    lcfg = O.LmCfg(hidden=4096, layers=32, heads=32, kv_heads=8, mlp=14336, vocab=32000, eps=1e-5, rope_theta=1e6)
    Wv, Wc = O.make_vit_weights(vcfg, 101), conn_gate_weights(ccfg, gcfg, 102)
    from streammind_amd.native import NativeModel
    m = NativeModel(path_config(vcfg, ccfg, gcfg, lcfg, max_frames_per_call=28))
    for k, v in Wv.items():
        m.load_tensor("model.vision_tower.vision_tower.vision_model." + k, v.to(torch.bfloat16) if v.dim() >= 2 else v)
    for k, v in Wc.items():
        m.load_tensor("model.mm_projector." + k, v.to(torch.bfloat16) if v.dim() >= 2 and "conv1d" not in k and "A_log" not in k else v)
    Wl = _LazyF32()
    _seeded_llm_into(m, lcfg, 2025, Wl)
    assert m.missing() == [], m.missing()
    m.finalize()
    torch.cuda.empty_cache()
    torch.set_num_threads(max(16, torch.get_num_threads()))
    frames = O.synthetic_frames(56, 336, seed=58, scene_len=7)
    fg = frames.cuda()
    # ---- the oracle's perception of the whole clip (what the reference recomputes at every frame)
    feats = torch.cat([O.vit_features(O.preprocess_frames(frames[i:i + 4]), Wv, vcfg, O.MIXED) for i in range(0, 56, 4)])
    tok_ref = O.connector_scan(O.pool_patches(feats), Wc, ccfg)
    lg_ref = O.gate_logits_shortcut(tok_ref, Wc, gcfg)
    del feats
    table = Wl["model.embed_tokens.weight"]
    g = torch.Generator().manual_seed(7)
    t1, t2, t3 = (torch.randint(3, lcfg.vocab, (n,), generator=g) for n in (34, 8, 12))
    s = m.open_stream(max_frames=64, max_seq=768)
    # ---- tick 1: frames 0..27, fire
    lg_a, dc_a = s.push_frames(fg[:28])
    ids1 = torch.cat([t1, -(torch.arange(28) + 1), t2]).to(torch.int32)
    s.prefill(ids1.cuda())
    emb1 = torch.cat([table[t1], tok_ref[:28], table[t2]])
    reply1, f1 = _decode_check(s, Wl, lcfg, emb1, 256, 3e-2, "ViT-L + Mistral-7B, fire 1")
    assert s.kv_len == 70 + 256
    # ---- tick 2: frames 28..55, fire; only the new positions are prefilled
    lg_b, dc_b = s.push_frames(fg[28:])
    ids2 = torch.cat([-(torch.arange(28, 56) + 1), t3]).to(torch.int32)
    s.prefill(ids2.cuda())
    emb2 = torch.cat([emb1, table[torch.tensor(reply1)], tok_ref[28:], table[t3]])
    _, f2 = _decode_check(s, Wl, lcfg, emb2, 256, 3e-2, "ViT-L + Mistral-7B, fire 2 (prefix reuse vs from-scratch)")
    assert s.kv_len == 70 + 256 + 40 + 256
    lg, dc = torch.cat([lg_a, lg_b]).cpu(), torch.cat([dc_a, dc_b]).cpu()
    dl, dt = maxdiff(lg, lg_ref), maxdiff(s.tokens(), tok_ref)
    print(f"56 frames: gate logits max|diff| vs the bf16-mode oracle {dl:.3e}; frame tokens {dt:.3e} (max |token| {float(tok_ref.abs().max()):.2f}); "
          f"near-tie id choices {f1} + {f2} of 2 x 256")
    assert dl < 2e-3 and dt < 3e-3
    for j in range(56):
        if abs(float(lg_ref[j, 1] - lg_ref[j, 0])) > 4e-3:
            assert int(dc[j]) == O.gate_decision(lg_ref[j])


def _planted_llm(lcfg, seed, peak: float):
    """Mistral-shaped weights whose greedy continuation has a PLANTED top-2 margin: embed_tokens[v] = peak x u[succ(v)] + small noise and
    lm_head[w] = u[w] for random unit-ish directions u (4096-d, 32 000 of them: |cos| between two of them <~ 0.08) and a fixed
    permutation succ; the 8 transformer layers are random layers whose OUTPUT projections (o_proj, down_proj) are scaled by 0.05, so
    every kernel of the decode path runs on ordinary values and each sub-layer perturbs the residual stream by a few per cent instead
    of burying it (unscaled random layers add ~16 vectors of norm 64 to a planted component of norm `peak`: with the round's first
    version of this helper -- peak 3, unscaled layers -- the oracle's own top-2 margin fell to 1e-2 .. 1e-3, inside the 16-bit
    error, and 0 flips was luck).  The final hidden state keeps a dominant component along u[succ(v)] and the logit of succ(v) leads
    the runner-up by a margin far above the 16-bit error -- the situation of a confident trained model, where 'identical token ids'
    is a property a test can demand of EVERY step; the test asserts that margin."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    d, V = lcfg.hidden, lcfg.vocab
    u = torch.randn(V, d, generator=g, device="cuda") * d ** -0.5
    succ = torch.randperm(V, generator=g, device="cuda")
    W = {}
    for name, shp in O.lm_weight_shapes(lcfg, "", True).items():
        if name.endswith("embed_tokens.weight"):
            w = peak * u[succ] + 0.05 * torch.randn(V, d, generator=g, device="cuda")
        elif name.endswith("lm_head.weight"):
            w = u.clone()
        elif "layernorm" in name or name.endswith("model.norm.weight"):
            w = 1.0 + 0.1 * torch.randn(*shp, generator=g, device="cuda")
        else:
            w = torch.randn(*shp, generator=g, device="cuda") * shp[-1] ** -0.5
            if name.endswith(("o_proj.weight", "down_proj.weight")):
                w = w * 0.05
        W[name] = w.to(torch.bfloat16)
    return W, succ.cpu()


@pytest.mark.parametrize("llm_fp16", [False, True])
def test_256_greedy_ids_are_equal_on_planted_margins_and_flips_counted_on_random_weights(llm_fp16):
    """VERDICT r3 weak #3.  Mistral-7B widths (4096 / 32 q / 8 kv heads x 128 / MLP 14336 / vocab 32 000), 8 distinct layers, in
    bf16 and in llm_fp16 (the mode load_pretrained_model picks for the reference's fp16 checkpoints):
      * PLANTED margins (_planted_llm): after a 40-token prefill, all 256 greedy ids must be EQUAL to the oracle's greedy ids --
        no near-tie allowance (and the oracle's smallest top-2 margin over the 256 steps is printed: it must dwarf the tolerance);
      * RANDOM weights (no structure, the worst case for greedy decoding: the top-2 gap of 32 000 i.i.d. logits is routinely
        below any 16-bit error): 256 steps, logits within tolerance at every step, flips per 256 REPORTED for the mode."""
    lcfg = O.LmCfg(hidden=4096, layers=8, heads=32, kv_heads=8, mlp=14336, vocab=32000, eps=1e-5, rope_theta=1e6)
    vcfg = O.VitCfg(image_size=28, patch=14, hidden=1024, heads=16, mlp=64, layers=2)
    ccfg, gcfg = O.ConnCfg(), O.LmCfg.gate(layers=1)
    Wv, Wc = O.make_vit_weights(vcfg, 1), conn_gate_weights(ccfg, gcfg, 2)
    prec = O.MIXED_F16 if llm_fp16 else O.MIXED
    rel = 4e-3 if llm_fp16 else 3e-2
    torch.set_num_threads(max(16, torch.get_num_threads()))
    from streammind_amd.native import NativeModel

    def model_with(Wdev):
        m = NativeModel(path_config(vcfg, ccfg, gcfg, lcfg, llm_fp16=llm_fp16))
        for k, v in Wv.items():
            m.load_tensor("model.vision_tower.vision_tower.vision_model." + k, v.to(torch.bfloat16) if v.dim() >= 2 else v)
        for k, v in Wc.items():
            m.load_tensor("model.mm_projector." + k, v.to(torch.bfloat16) if v.dim() >= 2 and "conv1d" not in k and "A_log" not in k else v)
        Wl = _LazyF32()
        for name, w in Wdev.items():
            m.load_tensor(name, w.float() if w.dim() == 1 else w)      # bf16-exact values: exact in fp16 too (llm_fp16 stores fp16)
            Wl[name] = w.cpu()
        assert m.missing() == [], m.missing()
        m.finalize()
        return m, Wl

    g = torch.Generator().manual_seed(5)
    text = torch.randint(3, lcfg.vocab, (40,), generator=g)
    # ---- planted margins: every id equal
    Wdev, succ = _planted_llm(lcfg, 808, peak=16.0)
    m, Wl = model_with(Wdev)
    del Wdev
    torch.cuda.empty_cache()
    s = m.open_stream(max_frames=8, max_seq=320)
    s.prefill(text.to(torch.int32).cuda())
    emb = Wl["model.embed_tokens.weight"][text]
    ids, flips = _decode_check(s, Wl, lcfg, emb, 256, rel, f"planted margins, {'fp16' if llm_fp16 else 'bf16'} operands", prec)
    # flips == 0 IS id equality with the oracle's own greedy run: step j's oracle logits are those of the shared prefix ids[:j], so
    # by induction the oracle's greedy loop emits exactly `ids` (and its 256 step-by-step passes need not be paid for)
    assert flips == 0
    assert _decode_check.last_min_margin > 4 * _decode_check.last_tol, "the planted margin must dwarf the logit tolerance at EVERY step"
    chain = sum(1 for a, b in zip(ids[:-1], ids[1:]) if int(succ[a]) == b)
    print(f"  planted chain followed on {chain}/255 transitions")
    assert chain == 255
    del m, s, Wl
    torch.cuda.empty_cache()
    # ---- random weights: flips per 256 reported
    m = NativeModel(path_config(vcfg, ccfg, gcfg, lcfg, llm_fp16=llm_fp16))
    for k, v in Wv.items():
        m.load_tensor("model.vision_tower.vision_tower.vision_model." + k, v.to(torch.bfloat16) if v.dim() >= 2 else v)
    for k, v in Wc.items():
        m.load_tensor("model.mm_projector." + k, v.to(torch.bfloat16) if v.dim() >= 2 and "conv1d" not in k and "A_log" not in k else v)
    Wl = _LazyF32()
    _seeded_llm_into(m, lcfg, 909, Wl)
    m.finalize()
    s = m.open_stream(max_frames=8, max_seq=320)
    s.prefill(text.to(torch.int32).cuda())
    _, flips = _decode_check(s, Wl, lcfg, Wl["model.embed_tokens.weight"][text], 256, rel,
                             f"random weights, {'fp16' if llm_fp16 else 'bf16'} operands", prec)
    print(f"  FLIPS PER 256 (random weights, {'llm_fp16' if llm_fp16 else 'bf16'}): {flips}")
