"""mimic3_amd — an MI355X-native VITS inference engine behind the Mimic 3
``onnxruntime.InferenceSession.run`` boundary (``mimic3_tts/voice.py:230``).

Only the hot path lives here: the native library (``csrc/`` → ``libmi355vits.so``,
C ABI in ``include/mi355vits.h``) and the host-side mirror of the reference
interface (``session.InferenceSession``).  See DESIGN.md.
"""
from .config import VitsConfig  # noqa: F401

__version__ = "0.1.0"
