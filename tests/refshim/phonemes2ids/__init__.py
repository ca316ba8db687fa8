"""Stand-in for phonemes2ids (tests only).  Semantics taken from how Mimic 3 calls it (voice.py:126-152) and from
PhonemesConfig (config.py:146-191): ids of the phonemes of each word, the `blank` symbol between words and/or
tokens and at both ends, optional bos/eos, phonemes without an id dropped (fail_on_missing=False)."""
import enum
import typing

from . import utils  # noqa: F401


class BlankBetween(str, enum.Enum):
    TOKENS = "tokens"
    WORDS = "words"
    TOKENS_AND_WORDS = "tokens_and_words"


def load_phoneme_ids(ids_file: typing.Iterable[str], separator: str = " ") -> typing.Dict[str, int]:
    """Lines of '<id><separator><phoneme>' (phonemes.txt)."""
    out = {}
    for line in ids_file:
        line = line.rstrip("\r\n")
        if not line or line.startswith("# "):
            continue
        idx, _, phoneme = line.partition(separator)
        out[phoneme] = int(idx)
    return out


def phonemes2ids(word_phonemes, phoneme_to_id, pad=None, bos=None, eos=None, auto_bos_eos=False, blank=None,
                 blank_word=None, blank_between=BlankBetween.WORDS, blank_at_start=True, blank_at_end=True,
                 simple_punctuation=True, punctuation_map=None, separate=None, separate_graphemes=False,
                 separate_tones=False, tone_before=False, phoneme_map=None, fail_on_missing=True):
    if not isinstance(blank_between, BlankBetween):
        blank_between = BlankBetween(blank_between)
    blank_id = phoneme_to_id.get(blank) if blank is not None else None
    blank_word_id = phoneme_to_id.get(blank_word) if blank_word is not None else blank_id
    ids: typing.List[int] = []
    if auto_bos_eos and bos is not None and bos in phoneme_to_id:
        ids.append(phoneme_to_id[bos])
    if blank_id is not None and blank_at_start:
        ids.append(blank_id)
    words = [w for w in word_phonemes if w]
    for wi, word in enumerate(words):
        expanded = []
        for p in word:
            if punctuation_map and p in punctuation_map:
                p = punctuation_map[p]
            if phoneme_map and p in phoneme_map:
                expanded.extend(phoneme_map[p])
            else:
                expanded.append(p)
        for pi, p in enumerate(expanded):
            pid = phoneme_to_id.get(p)
            if pid is None:
                if fail_on_missing:
                    raise KeyError(p)
                continue
            ids.append(pid)
            last_token = pi == len(expanded) - 1
            if blank_id is not None and blank_between in (BlankBetween.TOKENS, BlankBetween.TOKENS_AND_WORDS) and not last_token:
                ids.append(blank_id)
        if wi < len(words) - 1 and blank_word_id is not None and blank_between in (BlankBetween.WORDS, BlankBetween.TOKENS_AND_WORDS):
            ids.append(blank_word_id)
    if blank_id is not None and blank_at_end:
        ids.append(blank_id)
    if auto_bos_eos and eos is not None and eos in phoneme_to_id:
        ids.append(phoneme_to_id[eos])
    return ids
