"""
Abstract interface of a quantum error correcting code, without pyQuil.

Mirrors the reference's qecc.py: the abstract QECC (qecc.py:51-64: n, k, t), so that CSSCode subclasses the same
base, and CodeBlock (qecc.py:14-49) on the pyQuil-free instruction tuples of quantum_css_codes_amd.quil.
"""
import abc

from .quil import Program, gates


class CodeBlock(object):
    """The physical qubits of one logical qubit together with the registers of its known X and Z errors
    (qecc.py:14-33); every error-correction round updates the registers instead of touching the qubits."""

    def __init__(self, qubits, x_errors, z_errors):
        n = len(qubits)
        if len(x_errors) != n:
            raise ValueError("x_errors is of incorrect size")
        if len(z_errors) != n:
            raise ValueError("z_errors is of incorrect size")
        self.n, self.qubits, self.x_errors, self.z_errors = n, qubits, x_errors, z_errors

    def reset(self, prog):
        """qecc.py:35-49: measure every qubit into its x_errors bit, flip the ones that read 1, clear both registers."""
        prog += (gates.MEASURE(self.qubits[i], self.x_errors[i]) for i in range(self.n))
        for i in range(self.n):
            prog.if_then(self.x_errors[i], Program(gates.X(self.qubits[i])))
            prog += gates.MOVE(self.x_errors[i], 0)
            prog += gates.MOVE(self.z_errors[i], 0)


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
