"""Plain stderr logger standing in for the reference's tqdmlogger wrapper (src/monitor/logger.py)."""
import os
import sys

QUIET = bool(int(os.environ.get("MASR_QUIET", "0")))


def _emit(tag, *args):
    if not QUIET:
        print(f"[{tag}]", *args, file=sys.stderr, flush=True)


def flush():
    sys.stderr.flush()


def log(*args, prefix=None, update=False):
    _emit(prefix or "log", *args)


def log_info(info, prefix):
    _emit(prefix, " ".join(f"{k}={float(v):.6f}" for k, v in info.items()))


def warning(msg):
    _emit("warning", msg)


def notice(msg):
    _emit("notice", msg)
