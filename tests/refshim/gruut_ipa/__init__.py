"""Stand-in for gruut_ipa (tests only): break symbols and grapheme splitting."""
import enum
import unicodedata


class IPA(str, enum.Enum):
    STRESS_PRIMARY = "ˈ"
    STRESS_SECONDARY = "ˌ"
    BREAK_SYLLABLE = "."
    BREAK_MINOR = "|"
    BREAK_MAJOR = "‖"
    BREAK_WORD = "#"
    TIE_ABOVE = "͡"
    TIE_BELOW = "͜"

    @staticmethod
    def graphemes(text):
        """Base characters with their combining marks; a tie joins the next base character too."""
        out = []
        tie = False
        for ch in unicodedata.normalize("NFC", text):
            if out and (unicodedata.combining(ch) or tie):
                out[-1] += ch
            else:
                out.append(ch)
            tie = ch in ("͡", "͜")
        return out
