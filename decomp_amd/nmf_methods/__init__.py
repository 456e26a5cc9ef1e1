from . import grads  # noqa: F401
