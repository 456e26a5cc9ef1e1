from . import assertion, dtype, exceptions  # noqa: F401
