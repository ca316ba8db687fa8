"""Test-only stand-in for `quart_cors`: CORS headers are irrelevant to the tests."""


def cors(app, **kwargs):
    return app
