from . import eigen, linalg  # noqa: F401
