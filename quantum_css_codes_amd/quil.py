"""
A pyQuil-free statement of the small part of pyQuil the reference's code generators use.

The reference emits its fault-tolerant programs through pyQuil 2.8 (`Program`, `pyquil.gates`, `MemoryReference`,
`QubitPlaceholder`, `Program.if_then / while_do`, `address_qubits`; quil_classical.py:5-7, css_code.py:7-11, ftqc.py:9-36),
which is not installable offline and absent on the GPU box.  This module keeps the instruction stream the reference builds --
same instructions, same order, same Quil text -- as plain tuples, so that `quil_classical.py`, `qecc.py`, the emitters of
`css_code.py` and `ftqc.py` can be mirrored line by line and checked by execution (oracle/quil_sim.py) instead of by a QVM.

An instruction is a tuple whose first element is the opcode:
    ("GATE", name, (qubits...))            ("MEASURE", qubit, MemoryReference or None)
    ("MOVE" | "AND" | "XOR" | "IOR" | "ADD" | "CONVERT", dst, src)      src: MemoryReference or int
    ("NOT", dst)                           ("GE", dst, a, b)
    ("LABEL", name)  ("JUMP", name)  ("JUMP-WHEN", name, ref)  ("JUMP-UNLESS", name, ref)
    ("DECLARE", name, type, size)  ("HALT",)  ("PRAGMA", text)
"""
import itertools


class MemoryReference(object):
    """Classical memory address `name[offset]` (pyquil.quilatom.MemoryReference).  Indexing is allowed off the base
    reference only, as in pyQuil."""

    def __init__(self, name, offset=0, declared_size=None):
        self.name, self.offset, self.declared_size = name, int(offset), declared_size

    def __getitem__(self, offset):
        if self.offset != 0:
            raise ValueError("Please only index off of the base MemoryReference (offset = 0)")
        if self.declared_size is not None and offset >= self.declared_size:
            raise IndexError("MemoryReference index out of range")
        return MemoryReference(self.name, int(offset))

    def out(self):
        return "%s[%d]" % (self.name, self.offset)

    def __eq__(self, other):
        return isinstance(other, MemoryReference) and (self.name, self.offset) == (other.name, other.offset)

    def __hash__(self):
        return hash((self.name, self.offset))

    def __repr__(self):
        return "<MRef %s>" % self.out()


class QubitPlaceholder(object):
    """A qubit without an index yet (pyquil.quilatom.QubitPlaceholder); `address_qubits` assigns one."""
    _ids = itertools.count()

    def __init__(self):
        self.id = next(QubitPlaceholder._ids)

    def __repr__(self):
        return "<QubitPlaceholder %d>" % self.id


def _fmt(arg):
    if isinstance(arg, MemoryReference):
        return arg.out()
    if isinstance(arg, QubitPlaceholder):
        return "{q%d}" % arg.id
    return str(arg)


class Program(object):
    """An instruction list with pyQuil's building conveniences: `+=` takes an instruction, a Program or any iterable of
    them; declare / if_then / while_do as in pyquil.quil.Program."""
    _labels = itertools.count(1)

    def __init__(self, *items):
        self.instructions = []
        for item in items:
            self += item

    def __iadd__(self, other):
        if isinstance(other, Program):
            self.instructions.extend(other.instructions)
        elif isinstance(other, tuple) and other and isinstance(other[0], str):
            self.instructions.append(other)
        elif other is not None:
            for item in other:
                self += item
        return self

    def inst(self, *items):
        for item in items:
            self += item
        return self

    append = inst

    def __len__(self):
        return len(self.instructions)

    def __iter__(self):
        return iter(self.instructions)

    def declare(self, name, memory_type='BIT', memory_size=1):
        self.instructions.append(("DECLARE", name, memory_type, int(memory_size)))
        return MemoryReference(name, 0, declared_size=int(memory_size))

    @staticmethod
    def _fresh(prefix):
        return "%s%d" % (prefix, next(Program._labels))

    def if_then(self, classical_reg, if_program, else_program=None):
        """pyQuil's layout: JUMP-WHEN @THEN reg; else branch; JUMP @END; LABEL @THEN; if branch; LABEL @END."""
        label_then, label_end = self._fresh("THEN"), self._fresh("END")
        self.instructions.append(("JUMP-WHEN", label_then, classical_reg))
        self += else_program if else_program is not None else Program()
        self.instructions.append(("JUMP", label_end))
        self.instructions.append(("LABEL", label_then))
        self += if_program
        self.instructions.append(("LABEL", label_end))
        return self

    def while_do(self, classical_reg, q_program):
        """pyQuil's layout: LABEL @START; JUMP-UNLESS @END reg; body; JUMP @START; LABEL @END."""
        label_start, label_end = self._fresh("START"), self._fresh("END")
        self.instructions.append(("LABEL", label_start))
        self.instructions.append(("JUMP-UNLESS", label_end, classical_reg))
        self += q_program
        self.instructions.append(("JUMP", label_start))
        self.instructions.append(("LABEL", label_end))
        return self

    def get_qubits(self):
        found = []
        for inst in self.instructions:
            if inst[0] == "GATE":
                found.extend(inst[2])
            elif inst[0] == "MEASURE":
                found.append(inst[1])
        seen, out = set(), []
        for q in found:
            key = q.id if isinstance(q, QubitPlaceholder) else ("i", q)
            if key not in seen:
                seen.add(key)
                out.append(q)
        return out

    def out(self):
        """Quil text, one instruction per line (the format pyQuil's Program.out() prints for these instructions)."""
        lines = []
        for inst in self.instructions:
            op = inst[0]
            if op == "GATE":
                lines.append("%s %s" % (inst[1], " ".join(_fmt(q) for q in inst[2])))
            elif op == "MEASURE":
                lines.append("MEASURE %s" % _fmt(inst[1]) + ("" if inst[2] is None else " " + _fmt(inst[2])))
            elif op == "DECLARE":
                lines.append("DECLARE %s %s[%d]" % (inst[1], inst[2], inst[3]))
            elif op == "LABEL":
                lines.append("LABEL @%s" % inst[1])
            elif op == "JUMP":
                lines.append("JUMP @%s" % inst[1])
            elif op in ("JUMP-WHEN", "JUMP-UNLESS"):
                lines.append("%s @%s %s" % (op, inst[1], _fmt(inst[2])))
            elif op == "PRAGMA":
                lines.append("PRAGMA %s" % inst[1])
            else:
                lines.append(" ".join([op] + [_fmt(a) for a in inst[1:]]))
        return "\n".join(lines) + ("\n" if lines else "")


def address_qubits(program):
    """pyquil.quil.address_qubits: placeholders get the lowest indices the program does not use yet, in order of first
    appearance.  Returns a new Program."""
    used = {q for q in program.get_qubits() if not isinstance(q, QubitPlaceholder)}
    fresh = (i for i in itertools.count() if i not in used)
    mapping = {}

    def addr(q):
        if isinstance(q, QubitPlaceholder):
            if q.id not in mapping:
                mapping[q.id] = next(fresh)
            return mapping[q.id]
        return q

    out = Program()
    for inst in program.instructions:
        if inst[0] == "GATE":
            out.instructions.append(("GATE", inst[1], tuple(addr(q) for q in inst[2])))
        elif inst[0] == "MEASURE":
            out.instructions.append(("MEASURE", addr(inst[1]), inst[2]))
        else:
            out.instructions.append(inst)
    return out


class _Gates(object):
    """pyquil.gates: the constructors the reference uses."""

    @staticmethod
    def _one(name):
        return lambda qubit: ("GATE", name, (qubit,))

    @staticmethod
    def _two(name):
        return lambda a, b: ("GATE", name, (a, b))

    def __init__(self):
        for name in ("I", "X", "Y", "Z", "H", "S"):
            setattr(self, name, self._one(name))
        for name in ("CNOT", "CZ"):
            setattr(self, name, self._two(name))
        self.QUANTUM_GATES = {name: getattr(self, name) for name in ("I", "X", "Y", "Z", "H", "S", "CNOT", "CZ")}

    @staticmethod
    def MEASURE(qubit, classical_reg):
        return ("MEASURE", qubit, classical_reg)

    @staticmethod
    def MOVE(dst, src):
        return ("MOVE", dst, src)

    @staticmethod
    def AND(dst, src):
        return ("AND", dst, src)

    @staticmethod
    def XOR(dst, src):
        return ("XOR", dst, src)

    @staticmethod
    def IOR(dst, src):
        return ("IOR", dst, src)

    @staticmethod
    def NOT(dst):
        return ("NOT", dst)

    @staticmethod
    def CONVERT(dst, src):
        return ("CONVERT", dst, src)

    @staticmethod
    def ADD(dst, src):
        return ("ADD", dst, src)

    @staticmethod
    def GE(dst, a, b):
        return ("GE", dst, a, b)


gates = _Gates()
