"""Stand-in for xdgenvpy (tests only)."""
import os


class XDG:
    @property
    def XDG_DATA_HOME(self):
        return os.environ.get("XDG_DATA_HOME") or os.path.join(os.path.expanduser("~"), ".local", "share")

    @property
    def XDG_DATA_DIRS(self):
        return ":".join([self.XDG_DATA_HOME, os.environ.get("XDG_DATA_DIRS") or "/usr/local/share:/usr/share"])
