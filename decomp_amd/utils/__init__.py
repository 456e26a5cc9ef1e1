from . import assertion, dtype, exceptions, data, normalize, cp_compat  # noqa: F401
