"""
Abstract interface of a quantum error correcting code, without pyQuil.

Mirrors the shape of the reference's qecc.py:51-64 (abstract n, k, t) so that CSSCode subclasses the
same base.  CodeBlock (qecc.py:14-49) wraps pyQuil qubit placeholders and Quil memory and belongs to
the Quil-emission layer, which is out of scope.
"""
import abc


class QECC(abc.ABC):
    """Abstract Quantum Error Correcting Code."""

    def __init__(self):
        pass

    @property
    @abc.abstractmethod
    def n(self):
        """Number of physical qubits per code block."""
        raise NotImplementedError()

    @property
    @abc.abstractmethod
    def k(self):
        """Number of logical qubits per code block."""
        raise NotImplementedError()

    @property
    @abc.abstractmethod
    def t(self):
        """Maximum number of errors per code block that can be corrected."""
        raise NotImplementedError()
