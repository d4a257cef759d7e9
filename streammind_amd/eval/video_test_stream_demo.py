"""eval/video_test_stream_demo.py of the reference: the same `model_init` (:42-63) and streaming `infer` (:66-125) as
eval/video_score_stream_demo.py, driven by its OWN main loop (:147-182): ONE video and one fixed instruction instead of the
score demo's dataset loader (`build_score_eval`, which lives in the reference's missing `data/` package), every sampled frame
handed to `infer` alone, and a line printed for every frame the gate fired on.

    run_inference_time_metric(args)      eval/video_test_stream_demo.py:147-182
    python -m videollama2.eval.video_test_stream_demo --model-path CKPT --video-path VIDEO [--cur_fps 2]      :184-212

What differs from the reference's text, and why: the video path is an argument (`--video-path`; upstream it is a literal under a
developer's home directory, :152), the two live `pdb.set_trace()` calls (:148-149) are absent, and the loop returns what it printed.
"""
from __future__ import annotations

import argparse

from .video_score_stream_demo import get_index_stream, infer, model_init, read_video_stream  # noqa: F401  (:42-145: the same functions)

# the instruction the reference's loop hands to every tick (:153); `infer` overwrites it with '<video>\n' (:92), as upstream
INSTRUCT = "Please describe the video content in detail based on the provided information."
VIDEO_ORI_FPS = 30            # :161 -- the frame filter's time base
PRINT_FPS = 25                # :180 -- the printed mm:ss divides by 25, not by VIDEO_ORI_FPS, upstream too


def _frame_passes(frame_id: int, cur_min: int, cur_sec: int) -> bool:
    """the filter of :163.  `cur_min` / `cur_sec` are never advanced upstream (both stay -1), so every sampled frame passes;
    kept as the reference spells it so that a caller who does advance them gets the reference's behaviour."""
    t = frame_id // VIDEO_ORI_FPS
    return (t // 60 == cur_min and t % 60 > cur_sec) or (t // 60 > cur_min)


def run_inference_time_metric(args, on_reply=print):
    """eval/video_test_stream_demo.py:147-182.  `args` carries model_path / model_base / model_name / cur_fps as upstream plus
    video_path; optional `args.model` = an already initialised (model, processor, tokenizer, version) tuple (tests, servers).
    Returns [(frame_id, line)] -- the lines the reference prints, in order."""
    from PIL import Image
    init = getattr(args, "model", None)
    model, processor, tokenizer, version = init if init is not None else model_init(args.model_path, model_base=args.model_base, model_name=args.model_name)
    video_frames, vr = read_video_stream(args.video_path, args.cur_fps)
    cur_min = cur_sec = -1
    prompt = None
    out = []
    for frame_id in video_frames:
        frame_id = int(frame_id)
        if not _frame_passes(frame_id, cur_min, cur_sec):
            continue
        fr = vr[frame_id]
        img = Image.fromarray(fr.asnumpy() if hasattr(fr, "asnumpy") else fr)
        images_group = [img]
        video_frame = processor(images_group, num_frames=len(images_group))
        pred, prompt = infer(video=video_frame, instruct=INSTRUCT, model=model, tokenizer=tokenizer, do_sample=False, version=version,
                             score_video=True, prompt=prompt, **({"max_new_tokens": args.max_new_tokens} if getattr(args, "max_new_tokens", None) else {}))
        if pred is not None and pred != "":
            line = "The content of the video until {}:{}  is: {}:".format(frame_id // PRINT_FPS // 60, frame_id // PRINT_FPS % 60, pred)
            out.append((frame_id, line))
            if on_reply is not None:
                on_reply(line)
    return out


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--model-path", required=True)       # (upstream: a default under a developer's home directory)
    parser.add_argument("--model-name", default=None)
    parser.add_argument("--model-base", default=None)
    parser.add_argument("--eval-cls", default=None)
    parser.add_argument("--eval-caption", default=None)
    parser.add_argument("--batch-size", type=int, default=1)
    parser.add_argument("--num-workers", type=int, default=8)
    parser.add_argument("--cur_fps", type=int, default=2)
    parser.add_argument("--video-path", required=True)
    parser.add_argument("--max-new-tokens", type=int, default=None)
    args = parser.parse_args(argv)
    if args.model_name is None:
        args.model_name = "VideoLLaMA2-7B"
    return run_inference_time_metric(args)


if __name__ == "__main__":
    main()
