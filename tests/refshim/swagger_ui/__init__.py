"""Test-only stand-in for `swagger_ui`: no OpenAPI page in the tests."""


def api_doc(app, **kwargs):
    return None
