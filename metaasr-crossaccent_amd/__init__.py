"""MI355X-native meta-ASR training path (host mirror of the reference's Interface x Trainer contract).

The directory name carries a hyphen, so import it with
    import importlib; masr = importlib.import_module("metaasr-crossaccent_amd")
or through the `masr_amd` alias module at the repo root.
"""
from . import _cabi  # noqa: F401

__all__ = ["_cabi"]
