"""Stand-in for phonemes2ids.utils (tests only)."""


def load_phoneme_map(map_file, separator=" "):
    """Lines of '<phoneme> <phoneme> <phoneme> ...' -> {first: [rest]}."""
    out = {}
    for line in map_file:
        parts = line.strip().split(separator)
        if len(parts) >= 2:
            out[parts[0]] = parts[1:]
    return out
