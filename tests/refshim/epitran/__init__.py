"""Empty stand-in for epitran (tests only): imported at module top by mimic3_tts/voice.py, not used by SymbolsVoice."""


def _unavailable(*a, **k):
    raise RuntimeError("epitran is not installed in this environment (tests/refshim stand-in)")


sentences = Phonemizer = Epitran = _unavailable
