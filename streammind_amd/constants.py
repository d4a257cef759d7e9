"""Sentinel ids and defaults of the streaming path (mirror of streammind/constants.py:6-7,13-31)."""
NUM_FRAMES = 8
NUM_FRAMES_PER_SECOND = 1
MAX_FRAMES = 320000
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
MMODAL_TOKEN_INDEX = {"IMAGE": -200, "VIDEO": -201, "AUDIO": -202}
MMODAL_INDEX_TOKEN = {v: k for k, v in MMODAL_TOKEN_INDEX.items()}
DEFAULT_MMODAL_TOKEN = {"IMAGE": "<image>", "VIDEO": "<video>", "AUDIO": "<audio>"}
