"""Empty stand-in for gruut (tests only): imported at module top by mimic3_tts/voice.py, not used by SymbolsVoice."""


def _unavailable(*a, **k):
    raise RuntimeError("gruut is not installed in this environment (tests/refshim stand-in)")


sentences = Phonemizer = Epitran = _unavailable
