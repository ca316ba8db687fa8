"""What happens to the int16 audio after the hot path, restated so that a batch can be finished in one place
(SURVEY.md §8f N4).  The reference does these steps one sentence at a time on the host:

* volume — ``audioop.mul(audio_bytes, 2, settings.volume / 100)`` (``mimic3_tts/tts.py:542-543``): fused into the engine's
  int16 kernel (``InferenceSession.run_pcm16(feed, volume=...)``); ``apply_volume`` below is the same arithmetic in numpy
  for callers that already hold PCM;
* breaks — ``add_break``: ``int(ms / 1000 * sample_rate)`` zero samples (``tts.py:452-465``);
* WAV framing — ``wave.open`` around the concatenated bytes (``opentts_abc/__init__.py:117-127``).
"""
from __future__ import annotations

import io
import struct
from typing import Iterable, Optional, Sequence, Union

import numpy as np


def apply_volume(pcm: np.ndarray, volume_percent: float) -> np.ndarray:
    """``audioop.mul(pcm.tobytes(), 2, volume_percent / 100)`` as int16 array: double product, clipped to
    [-32768, 32767], rounded toward minus infinity (CPython ``Modules/audioop.c``, ``fbound``)."""
    pcm = np.asarray(pcm, dtype=np.int16)
    if float(volume_percent) == 100.0:
        return pcm.copy()
    d = pcm.astype(np.float64) * (float(volume_percent) / 100.0)
    d = np.where(d > 32767.0, 32767.0, np.where(d < -32767.0, -32768.0, d))
    return np.floor(d).astype(np.int16)


def silence(ms: Union[int, float], sample_rate: int = 22050) -> np.ndarray:
    """``Mimic3TextToSpeechSystem.add_break``: 16-bit mono zeros, ``int(ms / 1000 * sample_rate)`` samples."""
    return np.zeros(int((ms / 1000.0) * sample_rate), dtype=np.int16)


def wav_bytes(chunks: Iterable[np.ndarray], sample_rate: int = 22050) -> bytes:
    """One RIFF/WAVE file (PCM 16-bit mono) around the concatenated chunks — what ``text_to_wav`` returns."""
    data = b"".join(np.ascontiguousarray(c, dtype="<i2").tobytes() for c in chunks)
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate,
                                                                                         sample_rate * 2, 2, 16)
    return hdr + b"data" + struct.pack("<I", len(data)) + data


def utterances_to_wav(pcm_rows: Sequence[np.ndarray], sample_rate: int = 22050, break_ms: Optional[float] = None) -> bytes:
    """A batch of synthesised sentences (rows of ``run_pcm16``) -> one WAV, optional break between sentences."""
    parts = []
    for i, row in enumerate(pcm_rows):
        if i and break_ms:
            parts.append(silence(break_ms, sample_rate))
        parts.append(row)
    return wav_bytes(parts, sample_rate)
