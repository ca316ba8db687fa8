"""Make ``import onnxruntime`` resolve to the MI355X engine for an unchanged Mimic 3.

``mimic3_tts/voice.py:31`` imports ``onnxruntime`` at module top and uses three names from it:
``InferenceSession`` (``:403``), ``SessionOptions`` (``:393``) and ``GraphOptimizationLevel`` (``:397-399``).
Call :func:`install` before importing ``mimic3_tts``, or put this package's
``onnxruntime`` stub directory on ``PYTHONPATH``.
"""
from __future__ import annotations

import sys
import types

from .session import GraphOptimizationLevel, InferenceSession, NodeArg, SessionOptions  # noqa: F401

__version__ = "1.0.0+mi355vits"


def get_available_providers():
    return ["MI355XExecutionProvider"]


def get_device():
    return "GPU"


def install(force: bool = False) -> types.ModuleType:
    """Register this module as ``onnxruntime`` in ``sys.modules``."""
    if "onnxruntime" in sys.modules and not force:
        existing = sys.modules["onnxruntime"]
        if getattr(existing, "InferenceSession", None) is InferenceSession:
            return existing
        raise RuntimeError("a different 'onnxruntime' is already imported; call install(force=True) to replace it")
    mod = sys.modules[__name__]
    sys.modules["onnxruntime"] = mod
    return mod
