"""Teacher-forced evaluation metrics: the per-video arithmetic of
eval/inference_video_ego4d_stream_parallel_new.py:143-359 on the outputs of `model(..., llm_eval=True)` /
`model(..., model_type="cls", data_type="eval")`.  Plain tensor bookkeeping on a few hundred numbers per video; the
logits come from the HIP path (sm_llm_forward_logits, the gate step).  Pinned by golden g16: the script's two loop bodies,
compiled from its own source and executed on synthetic outputs (the script itself needs the missing data/ package)."""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from .constants import IGNORE_INDEX


def llm_turn_metrics(logits: torch.Tensor, labels: torch.Tensor, eos_id: int = 2) -> Dict[str, object]:
    """:190-222 for ONE video.  logits fp32 [S, vocab] (output.logits[0]), labels [1, S] (the expanded labels the
    forward returns).  A turn ends at label == eos_id; inside a turn position t is scored against label t+1 where that
    label is not IGNORE_INDEX.  -> per-video means as the script accumulates them:
    lm_ppl (mean over turns of exp(CE)), lm_correctness (mean token accuracy), lm_correct_tokens, lm_tokens, pred_ids."""
    lab = labels.reshape(-1).to("cpu")
    turns = (lab == eos_id).nonzero(as_tuple=True)[0].tolist()
    start = [-1] + turns[:-1]
    ppls, corr, ncorr, ntok, preds = [], [], [], [], []
    if logits.is_cuda:
        # logits still in HBM (what `model(..., llm_eval=True)` returns): the per-row NLL and arg-max of EVERY position in one library call
        # (sm_cross_entropy on the labels shifted by one); what is left per turn is a mean over a few dozen numbers
        from . import native as _native
        S = lab.numel()
        shifted = torch.full((S,), IGNORE_INDEX, dtype=torch.int32)
        shifted[:-1] = lab[1:].to(torch.int32)
        lg2 = logits.reshape(S, -1)
        nll_d, am_d = _native.cross_entropy(lg2 if lg2.dtype == torch.float32 and lg2.stride(1) == 1 else lg2.float().contiguous(), shifted)
        nll, am = nll_d.cpu(), am_d.cpu().long()
    else:
        lg = logits.float()
    for a, b in zip(start, turns):
        tg = lab[a + 1:b + 1][1:]
        keep = tg != IGNORE_INDEX
        tg = tg[keep]
        if logits.is_cuda:
            rows = torch.arange(a + 1, b)[keep]                      # position t of the turn is scored against label t + 1
            ppls.append(nll[rows].mean().exp())
            pred = am[rows]
        else:                                                        # host tensors (the CPU tests of this arithmetic, golden g16)
            tl = lg[a + 1:b + 1][:-1][keep]
            ppls.append(torch.nn.functional.cross_entropy(tl, tg).exp())
            pred = tl.argmax(dim=-1)
        ok = (pred == tg).sum()
        preds.append(pred.tolist())
        ncorr.append(ok); ntok.append(tg.numel()); corr.append(ok / tg.numel())
    n = max(len(turns), 1)
    return {"lm_ppl": float(sum(ppls) / n), "lm_correctness": float(sum(corr) / n),
            "lm_correct_tokens": float(sum(ncorr) / n), "lm_tokens": float(sum(ntok) / n), "pred_ids": preds}


def relaxed_correct(eos_labels: torch.Tensor, pred_labels: torch.Tensor, N: int) -> torch.Tensor:
    """:128-138: position i counts as matched when its label occurs among the predictions of frames i-N .. i+N."""
    matches = torch.zeros_like(eos_labels, dtype=torch.bool)
    for i in range(len(eos_labels)):
        lo, hi = max(0, i - N), min(len(eos_labels), i + N + 1)
        if eos_labels[i] in pred_labels[lo:hi]:
            matches[i] = True
    return matches


def gate_metrics(logits: torch.Tensor, labels: torch.Tensor, tolerance_frames: int = 2) -> Dict[str, float]:
    """:263-345 for ONE video.  logits [T, 2, 2] and labels [T, 2] as `model(..., model_type="cls")` returns them; the
    script scores logits[..., :-1, :] against labels[..., 1:], cuts the frame sequence into turns (a turn ends at a
    respond frame, label 1; trailing silent frames after the last respond frame are dropped, as its cache logic does) and
    reports, with the script's own names:
      accuracy            relaxed (+-tolerance_frames) matches / scored frames
      true_positive_rate  1 - (silent frames predicted respond and not relaxed-matched) / silent frames
      true_negative_rate  1 - (respond frames predicted silent and not relaxed-matched) / respond frames
      time_diffs          per turn: (number of wrong frames) / 2
      time_total / correct_time_total   mean frames per turn / mean exactly-right frames per turn"""
    lg = logits[:, :-1, :].float().to("cpu")          # [T, 1, 2]
    lb = labels[:, 1:].to("cpu")                      # [T, 1]
    turn_lg: List[torch.Tensor] = []
    turn_lb: List[torch.Tensor] = []
    cache_lg: List[torch.Tensor] = []
    cache_lb: List[torch.Tensor] = []
    for i in range(lb.shape[0]):
        cache_lg.append(lg[i]); cache_lb.append(lb[i])
        if lb[i] != 0:
            turn_lg.append(torch.cat(cache_lg)); turn_lb.append(torch.cat(cache_lb))
            cache_lg, cache_lb = [], []
    if not turn_lb:
        return {"accuracy": float("nan"), "true_positive_rate": float("nan"), "true_negative_rate": float("nan"),
                "time_diffs": [], "time_total": float("nan"), "correct_time_total": float("nan")}
    flat_lg, flat_lb = torch.cat(turn_lg), torch.cat(turn_lb)
    pred = torch.softmax(flat_lg, dim=-1).argmax(dim=-1)
    rel = relaxed_correct(flat_lb, pred, tolerance_frames)
    acc = rel.sum().item() / (flat_lb.numel() + 1e-9)
    fp = (((flat_lb == 0) & (pred == 1)) & ~rel).sum().item()
    tpr = 1 - fp / ((flat_lb == 0).sum().item() + 1e-9)
    fn = (((flat_lb == 1) & (pred == 0)) & ~rel).sum().item()
    tnr = 1 - fn / ((flat_lb == 1).sum().item() + 1e-9)
    time_diffs, tot, cor = [], [], []
    for l, t in zip(turn_lg, turn_lb):
        wrong = l.argmax(dim=-1) != t
        time_diffs.append(float(wrong.sum()) / 2 if wrong.any() else 0.0)
        tot.append(t.numel()); cor.append(int((l.argmax(dim=-1) == t).sum()))
    return {"accuracy": acc, "true_positive_rate": tpr, "true_negative_rate": tnr, "time_diffs": time_diffs,
            "time_total": sum(tot) / len(tot), "correct_time_total": sum(cor) / len(cor)}
