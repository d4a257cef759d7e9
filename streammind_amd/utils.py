"""The small helpers the serving scripts import from `videollama2.utils` (streammind/utils.py:11-12,17-57,93-99,121-126):
the two user-facing message strings, a rotating-file logger, `disable_torch_init`, `pretty_print_semaphore`.  The reference's
stdout/stderr capture and its OpenAI moderation call are serving infrastructure outside the hot path and are not mirrored."""
from __future__ import annotations

import logging
import logging.handlers
import os

from .constants import LOGDIR

server_error_msg = "**NETWORK ERROR DUE TO HIGH TRAFFIC. PLEASE REGENERATE OR REFRESH THIS PAGE.**"
moderation_msg = "YOUR INPUT VIOLATES OUR CONTENT MODERATION GUIDELINES. PLEASE TRY AGAIN."

_handler = None


def build_logger(logger_name: str, logger_filename: str) -> logging.Logger:
    """daily-rotating file under LOGDIR + the root stream handler, same line format as the reference's"""
    global _handler
    fmt = logging.Formatter(fmt="%(asctime)s | %(levelname)s | %(name)s | %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    if not logging.getLogger().handlers:
        logging.basicConfig(level=logging.INFO)
    logging.getLogger().handlers[0].setFormatter(fmt)
    logger = logging.getLogger(logger_name)
    logger.setLevel(logging.INFO)
    if _handler is None:
        os.makedirs(LOGDIR, exist_ok=True)
        _handler = logging.handlers.TimedRotatingFileHandler(os.path.join(LOGDIR, logger_filename), when="D", utc=True, encoding="UTF-8")
        _handler.setFormatter(fmt)
    if _handler not in logger.handlers:
        logger.addHandler(_handler)
    return logger


def disable_torch_init() -> None:
    """the reference skips nn.Linear / nn.LayerNorm initialisation before from_pretrained; the native model allocates no
    torch modules, so there is nothing to skip"""


def violates_moderation(text: str) -> bool:
    """streammind/utils.py:102-118 asks the OpenAI moderation endpoint; there is no network on the serving box: never flags"""
    return False


def pretty_print_semaphore(semaphore) -> str:
    if semaphore is None:
        return "None"
    return f"Semaphore(value={semaphore._value}, locked={semaphore.locked()})"
