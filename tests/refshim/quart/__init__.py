"""Test-only stand-in for `quart` (not installed here): the part of its surface the reference's unmodified
``mimic3_http/app.py`` touches — an app object with ``route`` / ``errorhandler`` decorators, the ``request`` proxy, ``Response``,
``jsonify``, ``render_template``, ``send_from_directory`` — plus ``Quart.dispatch`` so that a test can call a route the way
the ASGI server would.  Never imported by the product."""
import contextvars
import inspect
import json
from pathlib import Path

_current = contextvars.ContextVar("quart_standin_request")


class _Args(dict):
    def get(self, key, default=None, type=None):  # noqa: A002 - quart's signature
        v = super().get(key, default)
        return type(v) if (type is not None and v is not None) else v


class Request:
    def __init__(self, method="GET", args=None, body=b"", content_type=None):
        self.method = method
        self.args = _Args(args or {})
        self.content_type = content_type
        self._body = body

    @property
    def data(self):
        async def _get():
            return self._body
        return _get()


class _RequestProxy:
    def __getattr__(self, name):
        return getattr(_current.get(), name)


request = _RequestProxy()


class Response:
    def __init__(self, response=None, status=200, mimetype=None, content_type=None, headers=None):
        self.response = response          # bytes / str / a (sync or async) iterator of byte chunks
        self.status_code = status
        self.mimetype = mimetype or content_type
        self.headers = dict(headers or {})

    async def get_data(self) -> bytes:
        r = self.response
        if r is None:
            return b""
        if isinstance(r, (bytes, bytearray)):
            return bytes(r)
        if isinstance(r, str):
            return r.encode()
        out = []
        if hasattr(r, "__aiter__"):
            async for chunk in r:
                out.append(bytes(chunk))
        else:
            for chunk in r:
                out.append(bytes(chunk))
        return b"".join(out)


def jsonify(obj):
    return Response(json.dumps(obj), mimetype="application/json")


async def render_template(name, **context):
    return f"<!-- template {name} {sorted(context)} -->"


async def send_from_directory(directory, filename):
    return Response(Path(directory, filename).read_bytes())


class Quart:
    def __init__(self, import_name, template_folder=None, **kwargs):
        self.import_name = import_name
        self.template_folder = template_folder
        self.config = {}
        self.secret_key = None
        self._routes = []
        self._error_handlers = []

    def route(self, rule, methods=("GET",), **kwargs):
        def deco(fn):
            self._routes.append((rule, tuple(methods), fn))
            return fn
        return deco

    def errorhandler(self, exc_type):
        def deco(fn):
            self._error_handlers.append((exc_type, fn))
            return fn
        return deco

    def _find(self, method, path):
        for rule, methods, fn in self._routes:
            if method not in methods:
                continue
            if "<" in rule:
                prefix = rule[: rule.index("<")]
                if path.startswith(prefix):
                    return fn, {"filename": path[len(prefix):]}
            elif rule == path:
                return fn, {}
        raise LookupError(f"no route for {method} {path}")

    async def dispatch(self, method, path, args=None, body=b"", content_type=None) -> Response:
        """Run the handler registered for (method, path) with a request context; exceptions go to the error handlers."""
        fn, kw = self._find(method, path)
        token = _current.set(Request(method, args, body, content_type))
        try:
            try:
                rv = fn(**kw)
                if inspect.isawaitable(rv):
                    rv = await rv
            except Exception as e:  # noqa: BLE001 - that is what an error handler is for
                for exc_type, handler in self._error_handlers:
                    if isinstance(e, exc_type):
                        rv = handler(e)
                        if inspect.isawaitable(rv):
                            rv = await rv
                        break
                else:
                    raise
            if isinstance(rv, tuple):
                return Response(rv[0], status=rv[1])
            if isinstance(rv, Response):
                return rv
            return Response(rv)
        finally:
            _current.reset(token)
