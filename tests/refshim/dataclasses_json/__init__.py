"""Stand-in for dataclasses_json (tests only): the three methods mimic3_tts/config.py uses."""
import dataclasses
import enum
import json
import typing


def _encode(v):
    if dataclasses.is_dataclass(v) and not isinstance(v, type):
        return {f.name: _encode(getattr(v, f.name)) for f in dataclasses.fields(v)}
    if isinstance(v, enum.Enum):
        return v.value
    if isinstance(v, dict):
        return {k: _encode(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_encode(x) for x in v]
    return v


def _decode(tp, v):
    if v is None:
        return None
    origin = typing.get_origin(tp)
    args = typing.get_args(tp)
    if origin is typing.Union:
        for a in args:
            if a is type(None):
                continue
            try:
                return _decode(a, v)
            except (ValueError, TypeError, KeyError):
                continue
        return v
    if isinstance(tp, type) and dataclasses.is_dataclass(tp):
        return _from_dict(tp, v)
    if isinstance(tp, type) and issubclass(tp, enum.Enum):
        return tp(v)
    if origin in (list, typing.List):
        return [_decode(args[0], x) for x in v] if args else list(v)
    if origin in (tuple, typing.Tuple):
        if args and args[-1] is Ellipsis:
            return tuple(_decode(args[0], x) for x in v)
        if args:
            return tuple(_decode(a, x) for a, x in zip(args, v))
        return tuple(v)
    if origin in (dict, typing.Dict):
        return {k: (_decode(args[1], x) if args else x) for k, x in v.items()}
    if tp in (int, float, str, bool):
        if tp is float and isinstance(v, int):
            return float(v)
        if not isinstance(v, tp):
            raise TypeError(f"{v!r} is not {tp}")
    return v


def _from_dict(cls, d):
    hints = typing.get_type_hints(cls)
    kwargs = {}
    for f in dataclasses.fields(cls):
        if f.name in d:
            kwargs[f.name] = _decode(hints.get(f.name, typing.Any), d[f.name])
    return cls(**kwargs)  # unknown keys are ignored, missing ones take the field default


class DataClassJsonMixin:
    def to_dict(self, encode_json=False):
        return _encode(self)

    def to_json(self, **kw):
        return json.dumps(self.to_dict(), **kw)

    @classmethod
    def from_dict(cls, d, infer_missing=False):
        return _from_dict(cls, d)

    @classmethod
    def from_json(cls, s, **kw):
        return _from_dict(cls, json.loads(s))
