"""eval/video_test_stream_demo.py of the reference: the same `model_init` (:42-63) and streaming `infer` (:66-125) as
eval/video_score_stream_demo.py, driven by a different main loop (:147-182)."""
from .video_score_stream_demo import get_index_stream, infer, model_init, read_video_stream, run_stream  # noqa: F401
