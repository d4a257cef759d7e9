"""Gate logits of a 26-row pass on fp8 weights: one 26-row pass (17..32-row fp8 kernel) against two 13-row passes (<= 16-row kernels) and the oracle (diagnostic)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import streammind_oracle as O
from tests.util_models import build_native, conn_gate_weights
from tests.test_gpu_path import TV, TC, TG, TL
Wv = O.make_vit_weights(TV, 41); Wc = conn_gate_weights(TC, TG, 86); Wl = O.make_lm_weights(TL, 44)
Wc8 = {k: (O.fp8_quantize_rows(v)[0] if (k.startswith("cls_net.") and v.dim() == 2 and "embed_tokens" not in k) else v) for k, v in Wc.items()}
pooled = torch.randn(26, TC.mm_hidden, generator=torch.Generator().manual_seed(5))
ref = O.gate_logits_shortcut(O.connector_scan(pooled, Wc8, TC), Wc8, TG)
ref_bf = O.gate_logits_shortcut(O.connector_scan(pooled, Wc, TC), Wc, TG)
for mode in (0, 1, 2):
    m = build_native(TV, TC, TG, Wv, Wc, TL, Wl, weights_fp8=mode)
    a = m.open_stream(max_frames=32, max_seq=64); b = m.open_stream(max_frames=32, max_seq=64)
    one, _ = a.push_pooled(pooled.cuda())
    two = torch.cat([b.push_pooled(pooled[:13].cuda())[0], b.push_pooled(pooled[13:].cuda())[0]])
    r = ref_bf if mode == 0 else ref
    print("mode", mode, "one pass vs oracle %.2e" % float((one.cpu() - r).abs().max()), "two passes vs oracle %.2e" % float((two.cpu() - r).abs().max()),
          "one vs two %.2e" % float((one - two).abs().max()))
