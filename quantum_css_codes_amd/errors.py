"""
Exception types of the package (same names and meaning as the reference's errors.py:5-9).
"""


class InvalidCodeError(Exception):
    """Raised where the reference raises it: css_code.py:75 and css_code.py:826."""


class UnsupportedGateError(Exception):
    """errors.py:8-9: raised by CSSCode.apply_gate for a logical gate that is neither a Pauli nor transversal."""


class UnsupportedQECCError(Exception):
    """ftqc.py:44 raises this name without defining it (a NameError in the reference); defined here."""


class UnsupportedProgramError(Exception):
    """ftqc.py:47,118 raise this name without defining it (a NameError in the reference); defined here."""
