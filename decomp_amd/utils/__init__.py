from . import assertion, dtype, exceptions, data  # noqa: F401
