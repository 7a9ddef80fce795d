"""
quantum_css_codes_amd -- MI355X-native GF(2) linear algebra and syndrome extraction behind the
bin_matrix / CSSCode surface of jimpo/quantum-css-codes.

    from quantum_css_codes_amd import bin_matrix, css_code
    from quantum_css_codes_amd.css_code import CSSCode

All GF(2) arithmetic runs in libgf2hip.so (hand-written HIP for gfx950, include/gf2hip.h); there is no
CPU fallback.
"""
from . import _native, bin_matrix, css_code, errors, ftqc, montecarlo, qecc, quil, quil_classical  # noqa: F401
from .css_code import CSSCode  # noqa: F401
from .errors import InvalidCodeError, UnsupportedGateError  # noqa: F401

__all__ = ["bin_matrix", "css_code", "errors", "qecc", "montecarlo", "quil", "quil_classical", "ftqc", "CSSCode",
           "InvalidCodeError", "UnsupportedGateError"]
