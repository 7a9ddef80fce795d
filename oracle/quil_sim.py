"""
ORACLE -- test infrastructure only.  NOT part of the product.

Executes the instruction tuples that quantum_css_codes_amd.quil / quil_classical / css_emit / ftqc emit, standing in for the
QVM the reference's tests use (test/test_quil_classical.py:13, test/test_ftqc.py:153-156):

  * classical instructions (MOVE, AND, XOR, IOR, NOT, CONVERT, ADD, GE) on named registers, with Quil's semantics;
  * control flow (LABEL, JUMP, JUMP-WHEN, JUMP-UNLESS, HALT);
  * the Clifford gates the emitters produce (I, X, Y, Z, H, S, CNOT, CZ) and MEASURE, on a stabiliser tableau
    (Aaronson & Gottesman, "Improved simulation of stabilizer circuits", 2004: n destabilisers + n stabilisers, rows of x | z
    bits and a sign) -- every program the reference generates is a stabiliser circuit, so no state vector is needed.

Random measurement outcomes come from a seeded generator.  `faults` injects Pauli errors: a dict {step: [(pauli, qubit), ...]}
applied just before the instruction with that program counter is executed for the first time.
"""
import numpy as np


class Tableau(object):
    def __init__(self, n):
        self.n = n
        self.x = np.zeros((2 * n, n), dtype=np.uint8)
        self.z = np.zeros((2 * n, n), dtype=np.uint8)
        self.r = np.zeros(2 * n, dtype=np.uint8)
        self.x[np.arange(n), np.arange(n)] = 1              # destabilisers X_i
        self.z[np.arange(n, 2 * n), np.arange(n)] = 1       # stabilisers Z_i

    def h(self, a):
        self.r ^= self.x[:, a] & self.z[:, a]
        self.x[:, a], self.z[:, a] = self.z[:, a].copy(), self.x[:, a].copy()

    def s(self, a):
        self.r ^= self.x[:, a] & self.z[:, a]
        self.z[:, a] ^= self.x[:, a]

    def cnot(self, a, b):
        self.r ^= self.x[:, a] & self.z[:, b] & (self.x[:, b] ^ self.z[:, a] ^ 1)
        self.x[:, b] ^= self.x[:, a]
        self.z[:, a] ^= self.z[:, b]

    def pauli(self, name, a):
        if name in ("X", "Y"):
            self.r ^= self.z[:, a]                          # X anticommutes with rows that carry Z on a
        if name in ("Z", "Y"):
            self.r ^= self.x[:, a]

    def _g(self, x1, z1, x2, z2):
        """Exponent of i (0, 1, -1) picked up when the single-qubit Paulis (x1 z1) and (x2 z2) multiply."""
        x1, z1, x2, z2 = (v.astype(np.int8) for v in (x1, z1, x2, z2))
        out = np.zeros(x1.shape, dtype=np.int8)
        both = (x1 == 1) & (z1 == 1)
        only_x = (x1 == 1) & (z1 == 0)
        only_z = (x1 == 0) & (z1 == 1)
        out[both] = (z2 - x2)[both]
        out[only_x] = (z2 * (2 * x2 - 1))[only_x]
        out[only_z] = (x2 * (1 - 2 * z2))[only_z]
        return out

    def _rowsum(self, h, i):
        total = 2 * int(self.r[h]) + 2 * int(self.r[i]) + int(self._g(self.x[i], self.z[i], self.x[h], self.z[h]).sum())
        self.r[h] = 1 if total % 4 == 2 else 0
        self.x[h] ^= self.x[i]
        self.z[h] ^= self.z[i]

    def measure(self, a, rng):
        n = self.n
        hits = np.flatnonzero(self.x[n:, a])
        if hits.size:                                       # random outcome
            p = n + int(hits[0])
            for i in range(2 * n):
                if i != p and self.x[i, a]:
                    self._rowsum(i, p)
            self.x[p - n], self.z[p - n], self.r[p - n] = self.x[p].copy(), self.z[p].copy(), self.r[p]
            self.x[p] = 0
            self.z[p] = 0
            self.z[p, a] = 1
            self.r[p] = int(rng.integers(0, 2))
            return int(self.r[p])
        # determined outcome: accumulate the stabilisers whose destabiliser partners carry X on a
        sx, sz, sr = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8), 0
        for i in np.flatnonzero(self.x[:n, a]):
            total = 2 * sr + 2 * int(self.r[i + n]) + int(self._g(self.x[i + n], self.z[i + n], sx, sz).sum())
            sr = 1 if total % 4 == 2 else 0
            sx ^= self.x[i + n]
            sz ^= self.z[i + n]
        return int(sr)


def _value(memory, arg):
    if isinstance(arg, int):
        return arg
    return int(memory[arg.name][arg.offset])


def run(program, memory=None, seed=0, faults=None, max_steps=50_000_000):
    """Executes the program; returns the memory dict (register name -> int64 array).  Registers are created by DECLARE or
    taken from `memory`."""
    insts = program.instructions
    memory = {} if memory is None else {k: np.array(v, dtype=np.int64) for k, v in memory.items()}
    labels = {inst[1]: pc for pc, inst in enumerate(insts) if inst[0] == "LABEL"}
    qubits = [q for q in program.get_qubits()]
    if any(not isinstance(q, (int, np.integer)) for q in qubits):
        raise ValueError("address the qubits first (quil.address_qubits)")
    tab = Tableau(max(qubits) + 1) if qubits else None
    rng = np.random.default_rng(seed)
    faults = dict(faults or {})
    pc, steps = 0, 0
    while pc < len(insts):
        steps += 1
        if steps > max_steps:
            raise RuntimeError("program did not halt")
        for pauli, q in faults.pop(pc, ()):
            tab.pauli(pauli, q)
        inst = insts[pc]
        op = inst[0]
        pc += 1
        if op == "GATE":
            name, qs = inst[1], inst[2]
            if name == "I":
                pass
            elif name in ("X", "Y", "Z"):
                tab.pauli(name, qs[0])
            elif name == "H":
                tab.h(qs[0])
            elif name == "S":
                tab.s(qs[0])
            elif name == "CNOT":
                tab.cnot(qs[0], qs[1])
            elif name == "CZ":
                tab.h(qs[1])
                tab.cnot(qs[0], qs[1])
                tab.h(qs[1])
            else:
                raise ValueError("not a stabiliser gate: %s" % name)
        elif op == "MEASURE":
            bit = tab.measure(inst[1], rng)
            if inst[2] is not None:
                memory[inst[2].name][inst[2].offset] = bit
        elif op == "DECLARE":
            memory.setdefault(inst[1], np.zeros(inst[3], dtype=np.int64))
        elif op == "MOVE":
            memory[inst[1].name][inst[1].offset] = _value(memory, inst[2])
        elif op == "AND":
            memory[inst[1].name][inst[1].offset] &= _value(memory, inst[2])
        elif op == "XOR":
            memory[inst[1].name][inst[1].offset] ^= _value(memory, inst[2])
        elif op == "IOR":
            memory[inst[1].name][inst[1].offset] |= _value(memory, inst[2])
        elif op == "NOT":
            memory[inst[1].name][inst[1].offset] = 1 - (memory[inst[1].name][inst[1].offset] & 1)   # BIT registers
        elif op == "CONVERT":
            memory[inst[1].name][inst[1].offset] = _value(memory, inst[2])
        elif op == "ADD":
            memory[inst[1].name][inst[1].offset] += _value(memory, inst[2])
        elif op == "GE":
            memory[inst[1].name][inst[1].offset] = 1 if _value(memory, inst[2]) >= _value(memory, inst[3]) else 0
        elif op == "LABEL" or op == "PRAGMA":
            pass
        elif op == "JUMP":
            pc = labels[inst[1]]
        elif op == "JUMP-WHEN":
            if _value(memory, inst[2]):
                pc = labels[inst[1]]
        elif op == "JUMP-UNLESS":
            if not _value(memory, inst[2]):
                pc = labels[inst[1]]
        elif op == "HALT":
            break
        else:
            raise ValueError("unsupported instruction %r" % (inst,))
    return memory
