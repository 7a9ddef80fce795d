"""
Exception types of the package (same names and meaning as the reference's errors.py:5-9).
"""


class InvalidCodeError(Exception):
    """Raised where the reference raises it: css_code.py:75 and css_code.py:826."""


class UnsupportedGateError(Exception):
    """Kept for interface compatibility (errors.py:8-9); gate emission is outside this package."""
