"""Probe of the two fp8 modes on the tiny LLM: HIP logits against the oracle with / without per-row e4m3 activations (how far the
activation quantisation moves the logits, and how chaotic it is w.r.t. bf16-level noise).   gpurun -- python tools/fp8_mode_probe.py"""
import torch, sys
sys.path.insert(0, "/root/repo")
from oracle import streammind_oracle as O
from tests.util_models import build_native, conn_gate_weights, fp8_view
from oracle.make_golden import TINY_V as TV, TINY_C as TC, TINY_G as TG, TINY_L as TL
torch.set_grad_enabled(False)
md = lambda a, b: (a.float().cpu() - b.float().cpu()).abs().max().item()
Wv = O.make_vit_weights(TV, 41); Wc = conn_gate_weights(TC, TG, 86); Wl = O.make_lm_weights(TL, 44)
Wl8 = fp8_view(Wl)
for mode in (1, 2):
    m = build_native(TV, TC, TG, Wv, Wc, TL, Wl, weights_fp8=mode)
    g = torch.Generator().manual_seed(4)
    tok = torch.randn(6, TL.hidden, generator=g) * 0.5
    s = m.open_stream(max_frames=32, max_seq=128)
    s.write_tokens(0, tok.cuda())
    ids = torch.cat([torch.tensor([1, 7, 9]), -(torch.arange(6) + 1), torch.tensor([11, 12] * 9), torch.tensor([5, 33, 71])]).to(torch.int32)
    s.prefill(ids.cuda()); lg, _ = s.logits()
    table = Wl["model.embed_tokens.weight"]
    emb = torch.cat([table[[1, 7, 9]], tok, table[[11, 12] * 9], table[[5, 33, 71]]])
    a = O.lm_forward(emb, Wl8, TL, O.KVCache(), O.MIXED_FP8ACT)
    b = O.lm_forward(emb, Wl8, TL, O.KVCache(), O.MIXED)
    c = O.lm_forward(emb, Wl8, TL, O.KVCache(), O.FP32)
    print(f"mode {mode}: hip-fp8act {md(lg,a):.4f} hip-weightonly {md(lg,b):.4f} fp8act-weightonly {md(a,b):.4f} mixed-fp32 {md(b,c):.4f} |logits| {a.abs().max():.2f}")
    s.set_kv_len(0)
    al = s.forward_logits(ids.cuda()).cpu()
    ra = O.lm_forward(emb, Wl8, TL, O.KVCache(), O.MIXED_FP8ACT, last_only=False)
    rb = O.lm_forward(emb, Wl8, TL, O.KVCache(), O.MIXED, last_only=False)
    print(f"   all rows: hip-fp8act {md(al,ra):.4f} hip-weightonly {md(al,rb):.4f} fp8act-wo {md(ra,rb):.4f}; per-row hip-fp8act {[(round(x,3)) for x in (al-ra).abs().amax(1).tolist()][:12]}")
    rms = lambda t: t.float().pow(2).mean().sqrt().item()
    print(f"   rel rms: hip-fp8act {rms(al-ra)/rms(ra):.4f} hip-wo {rms(al-rb)/rms(rb):.4f} fp8act-wo {rms(ra-rb)/rms(rb):.4f}")
