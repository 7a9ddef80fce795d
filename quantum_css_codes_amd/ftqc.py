"""
Fault-tolerant rewriting of a logical program: ftqc.rewrite_program of the reference (ftqc.py:42-171)
on the pyQuil-free instruction tuples of quantum_css_codes_amd.quil.

Every logical qubit of the input becomes a CodeBlock of qecc.n physical qubits plus 2 n bits of known-error registers; two
ancilla blocks and the scratch registers are shared by all operations (ftqc.py:149-153 explains why).  Logical qubits are
encoded to |0>, every logical gate is followed by a round of error correction on every block, a logical MEASURE becomes
2t + 1 noisy measurements with error correction in between and a majority vote.

Differences from the reference, all at places where the reference cannot run as written:
  * ftqc.py:44,47,118 raise UnsupportedQECCError / UnsupportedProgramError, names it never defines (NameError); they are
    defined in errors.py and raised here.
  * ftqc.py:110 passes block.qubits where encode_zero takes a CodeBlock (AttributeError on RESET); the block is passed.
  * ftqc.py:159 initialises the error-correction scratch with `ancilla_1.qubits + ancilla_1.qubits`; kept as written (any
    qubit serves: the MEASUREs only initialise QVM memory).
"""
from .errors import UnsupportedProgramError, UnsupportedQECCError
from .qecc import CodeBlock
from .quil import Program, QubitPlaceholder, address_qubits, gates
from .quil_classical import MemoryChunk

CLASSICAL_OPS = frozenset(("MOVE", "AND", "XOR", "IOR", "NOT", "CONVERT", "ADD", "GE", "NEG", "SUB", "MUL", "DIV", "EXCHANGE",
                           "LOAD", "STORE", "EQ", "GT", "LE", "LT"))


def rewrite_program(raw_prog, qecc):
    """ftqc.py:42-120."""
    if qecc.k != 1:
        raise UnsupportedQECCError("code must have k = 1")
    if any(inst[0] == "DEFGATE" for inst in raw_prog.instructions):
        raise UnsupportedProgramError("does not support DEFGATE")
    raw_prog = address_qubits(raw_prog)
    new_prog = Program()
    logical_qubits = {index: new_logical_qubit(new_prog, qecc, "logical_qubit_{}".format(index))
                      for index in sorted(raw_prog.get_qubits())}
    ancilla_1 = new_logical_qubit(new_prog, qecc, "ancilla_1")
    ancilla_2 = new_logical_qubit(new_prog, qecc, "ancilla_2")

    scratch_size = max(qecc.n, qecc.measure_scratch_size)
    raw_scratch = new_prog.declare('scratch', 'BIT', scratch_size)
    scratch = MemoryChunk(raw_scratch, 0, raw_scratch.declared_size)
    _initialize_memory(new_prog, raw_scratch, ancilla_1.qubits + ancilla_2.qubits)

    raw_scratch_int = new_prog.declare('scratch_int', 'INTEGER', 2)
    scratch_int = MemoryChunk(raw_scratch_int, 0, raw_scratch_int.declared_size)
    _initialize_memory(new_prog, raw_scratch_int, ancilla_1.qubits + ancilla_2.qubits)

    perform_error_correction = _make_error_corrector(new_prog, qecc, ancilla_1, ancilla_2)

    for block in logical_qubits.values():
        qecc.encode_zero(new_prog, block, ancilla_1, scratch)

    for inst in raw_prog.instructions:
        op = inst[0]
        if op == "GATE":
            qecc.apply_gate(new_prog, inst[1], *[logical_qubits[index] for index in inst[2]])
            perform_error_correction(logical_qubits.values())          # after every logical gate
        elif op == "MEASURE":
            # shares the ancillas with the error correction (ftqc.py:87-88: qubits are scarce)
            for _ in qecc.measure(new_prog, logical_qubits[inst[1]], 0, inst[2], ancilla_1, ancilla_2, scratch, scratch_int):
                perform_error_correction(logical_qubits.values())
        elif op == "RESET-QUBIT":
            raise NotImplementedError("this instruction is not in the Quil spec")
        elif op == "LABEL":
            new_prog.inst(("LABEL", _mangle_label(inst[1])))
        elif op in ("JUMP-WHEN", "JUMP-UNLESS"):
            new_prog.inst((op, _mangle_label(inst[1]), inst[2]))
        elif op == "JUMP":
            new_prog.inst(("JUMP", _mangle_label(inst[1])))
        elif op == "HALT":
            new_prog.inst(inst)
        elif op == "WAIT":
            raise NotImplementedError()
        elif op == "RESET":
            for block in logical_qubits.values():
                qecc.encode_zero(new_prog, block, ancilla_1, scratch)
        elif op in ("DECLARE", "PRAGMA") or op in CLASSICAL_OPS:
            new_prog.inst(inst)
        else:
            raise UnsupportedProgramError("unsupported instruction: {}".format(inst))
    return address_qubits(new_prog)


def new_logical_qubit(prog, qecc, name):
    """ftqc.py:122-128: n fresh qubits and a 2 n-bit register split into the X and the Z error halves."""
    n = qecc.n
    raw_mem = prog.declare(name, 'BIT', 2 * n)
    mem = MemoryChunk(raw_mem, 0, raw_mem.declared_size)
    qubits = [QubitPlaceholder() for _ in range(n)]
    _initialize_memory(prog, raw_mem, qubits)
    return CodeBlock(qubits, mem[:n], mem[n:])


def _initialize_memory(prog, mem, qubits):
    """ftqc.py:138-145: the QVM wants a MEASURE into a register before anything can be MOVEd there."""
    prog += (gates.MEASURE(qubits[i % len(qubits)], mem[i]) for i in range(mem.declared_size))
    prog += (gates.MOVE(mem[i], 0) for i in range(mem.declared_size))


def _mangle_label(label):
    """ftqc.py:147-151."""
    return "NESTED_{}".format(label)


def _make_error_corrector(prog, qecc, ancilla_1, ancilla_2):
    """ftqc.py:153-171: one shared scratch register and ancilla pair for every round of error correction."""
    scratch_size = max(qecc.n, qecc.error_correct_scratch_size)
    raw_scratch = prog.declare('error_correct_scratch', 'BIT', scratch_size)
    scratch = MemoryChunk(raw_scratch, 0, raw_scratch.declared_size)
    _initialize_memory(prog, raw_scratch, ancilla_1.qubits + ancilla_1.qubits)

    def perform_error_correction(blocks):
        for block in blocks:
            qecc.error_correct(prog, block, ancilla_1, ancilla_2, scratch)

    return perform_error_correction
