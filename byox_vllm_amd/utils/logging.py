"""Package logger (same entry points as nanovllm/utils/logging.py:5-33)."""

import logging
import sys

_FORMAT = "%(asctime)s - %(filename)s:%(lineno)d - %(levelname)s - %(message)s"


def init_logger(name: str | None = None, level: int = logging.WARNING) -> logging.Logger:
    root = logging.getLogger()
    if not root.handlers:
        logging.basicConfig(level=level, format=_FORMAT, datefmt="%H:%M:%S", stream=sys.stdout)
    log = logging.getLogger(name)
    log.setLevel(level)
    return log


def set_logger_level(log: logging.Logger, level: int) -> None:
    log.setLevel(level)


def set_global_log_level(level: int) -> None:
    root = logging.getLogger()
    root.setLevel(level)
    for handler in root.handlers:
        handler.setLevel(level)
    logger.setLevel(level)


logger = init_logger(__name__.split(".")[0])
