"""Full-size perception parity probe (28 frames): gate logits / pooled features of the HIP path at 28 and 2 frames per call
against the oracle in fp32 and in the mode that rounds where the HIP path rounds (O.MIXED).  Prints per-frame errors.
    gpurun -- 'python tools/fullsize_parity_probe.py'"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import streammind_oracle as O                      # noqa: E402
from tests.util_models import build_native, conn_gate_weights  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(64)
vcfg, ccfg, gcfg = O.VitCfg(), O.ConnCfg(), O.LmCfg.gate()
Wv = O.make_vit_weights(vcfg, 101)
Wc = conn_gate_weights(ccfg, gcfg, 102)
F16 = os.environ.get("VIT_FP16", "0") == "1"                   # the fp16 tower (vit_fp16) instead of the bf16 one
m = build_native(vcfg, ccfg, gcfg, Wv, Wc, max_frames_per_call=28, vit_fp16=F16)
print(f"tower operands: {'fp16' if F16 else 'bf16'}; SM_VIT_LN_FOLD={os.environ.get('SM_VIT_LN_FOLD', '(default 1)')}")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
frames = O.synthetic_frames(n, 336, seed=56, scene_len=5)
fg = frames.cuda()


def run(per):
    s = m.open_stream(max_frames=64, max_seq=64)
    lg = torch.cat([s.push_frames(fg[i:i + per].contiguous())[0] for i in range(0, n, per)]).cpu()
    pooled = torch.cat([m.vit_encode(fg[i:i + per].contiguous()) for i in range(0, n, per)]).cpu()
    return lg, pooled, s.tokens().cpu()


lg28, p28, t28 = run(min(28, n))
lg2, p2, t2 = run(2)


def oracle(prec):
    feats = torch.cat([O.vit_features(O.preprocess_frames(frames[i:i + 4]), Wv, vcfg, prec) for i in range(0, n, 4)])
    pooled = O.pool_patches(feats)
    tok = O.connector_scan(pooled, Wc, ccfg)
    return pooled, tok, O.gate_logits_shortcut(tok, Wc, gcfg)


p32, t32, l32 = oracle(O.FP32)
if os.environ.get("PROBE_JSON") == "1":        # tests/test_gpu_path.py::test_bf16_tower_with_the_fold_forced_on: one JSON line, the fold's oracle only
    import json
    pfo, tfo, lfo = oracle(O.MIXED_F16_FOLD if F16 else O.MIXED_FOLD)
    md_ = lambda a, b: (a - b).abs().max().item()
    print(json.dumps({"fp16": F16, "fold_env": os.environ.get("SM_VIT_LN_FOLD"), "logits_gpu28_vs_fp32": md_(lg28, l32), "logits_gpu28_vs_fold_oracle": md_(lg28, lfo),
                      "logits_gpu2_vs_fp32": md_(lg2, l32), "logits_gpu28_vs_gpu2": md_(lg28, lg2), "pooled_gpu28_vs_fold_oracle": md_(p28, pfo), "pooled_max": p32.abs().max().item(),
                      "fold_oracle_vs_fp32": md_(lfo, l32)}))
    sys.exit(0)
pmx, tmx, lmx = oracle(O.MIXED_F16 if F16 else O.MIXED)
pfo, tfo, lfo = oracle(O.MIXED_F16_FOLD if F16 else O.MIXED_FOLD)          # the tower's LayerNorms folded into the neighbouring GEMMs (what the HIP path does at >= 21 frames per lane)
# connector + gate alone, fed with the GPU's own pooled features: isolates the ViT's contribution
l_from_gpu_pooled = O.gate_logits_shortcut(O.connector_scan(p28, Wc, ccfg), Wc, gcfg)
md = lambda a, b: (a - b).abs().max().item()
print(f"max |pooled| {p32.abs().max():.2f}  rms pooled {p32.pow(2).mean().sqrt():.3f}")
print(f"pooled: gpu28-fp32 {md(p28, p32):.3e}  gpu2-fp32 {md(p2, p32):.3e}  gpu28-gpu2 {md(p28, p2):.3e}  gpu28-mixed {md(p28, pmx):.3e}  mixed-fp32 {md(pmx, p32):.3e}")
print(f"pooled rms err: gpu28-fp32 {(p28 - p32).pow(2).mean().sqrt():.3e}  mixed-fp32 {(pmx - p32).pow(2).mean().sqrt():.3e}  gpu28-mixed {(p28 - pmx).pow(2).mean().sqrt():.3e}")
print(f"logits: gpu28-fp32 {md(lg28, l32):.3e}  gpu2-fp32 {md(lg2, l32):.3e}  gpu28-gpu2 {md(lg28, lg2):.3e}  gpu28-mixed {md(lg28, lmx):.3e}  mixed-fp32 {md(lmx, l32):.3e}")
print(f"logits: gpu28 vs oracle conn+gate on the GPU's pooled features {md(lg28, l_from_gpu_pooled):.3e}")
print(f"logits: gpu28-mixed_fold {md(lg28, lfo):.3e}  gpu2-mixed_fold {md(lg2, lfo):.3e}  mixed_fold-fp32 {md(lfo, l32):.3e}  mixed_fold-mixed {md(lfo, lmx):.3e};  pooled gpu28-mixed_fold {md(p28, pfo):.3e}")
print("per-frame |gpu28 - fp32| logits:", [f"{v:.1e}" for v in (lg28 - l32).abs().amax(1).tolist()])
print("per-frame |mixed - fp32| logits:", [f"{v:.1e}" for v in (lmx - l32).abs().amax(1).tolist()])
print("per-frame |gpu28 - mixed| logits:", [f"{v:.1e}" for v in (lg28 - lmx).abs().amax(1).tolist()])
print("oracle logits:", [[round(x, 3) for x in r] for r in l32.tolist()][:6], " margins min", float((l32[:, 1] - l32[:, 0]).abs().min()))
