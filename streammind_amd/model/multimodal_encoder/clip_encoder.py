"""streammind/model/multimodal_encoder/clip_encoder.py: module path of `CLIPVisionTower` (class in ../stream_model.py)"""
from ..stream_model import CLIPVisionTower  # noqa: F401
