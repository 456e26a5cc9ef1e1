"""Typed validation errors of the drop-in API.

Same names and base class as the reference's decomp/utils/exceptions.py:1-10, so
``except ShapeMismatchError`` written against deComP keeps working.
"""


class ShapeMismatchError(ValueError):
    """Two arrays whose shapes must agree do not."""


class DimInvalidError(ValueError):
    """An array has the wrong number of dimensions."""


class DtypeMismatchError(ValueError):
    """Arrays have different or unsupported dtypes."""
