"""
The pyQuil-free program emitters (quil_classical.py, qecc.CodeBlock, css_code.py:314-713, ftqc.py), checked by
execution: oracle/quil_sim.py stands in for the QVM of the reference's tests (a classical interpreter plus a stabiliser tableau).

CPU tests run the emitters on the oracle's NumPy CSSCode; the `-m gpu` tests run them on the product's GPU-built CSSCode and
compare the emitted classical decode with the GPU syndrome path on sampled errors.
"""
import numpy as np
import pytest

from oracle import cpu_ref, quil_sim
from quantum_css_codes_amd import css_emit, ftqc, quil_classical
from quantum_css_codes_amd.errors import UnsupportedGateError, UnsupportedQECCError
from quantum_css_codes_amd.qecc import CodeBlock
from quantum_css_codes_amd.quil import MemoryReference, Program, QubitPlaceholder, address_qubits, gates
from quantum_css_codes_amd.quil_classical import MemoryChunk


class OracleCode(object):
    """The oracle's NumPy CSSCode with the emitter methods of the product bound to it (no GPU needed)."""

    def __init__(self, h1, h2):
        self._code = cpu_ref.CSSCode(h1, h2)

    def __getattr__(self, name):
        return getattr(self._code, name)

    def encode_zero_gates(self):
        return cpu_ref.encode_zero_gates(self._code)

    def encode_plus_gates(self):
        return cpu_ref.encode_plus_gates(self._code)

    encode_scratch_size = property(css_emit.encode_scratch_size)
    error_correct_scratch_size = property(css_emit.error_correct_scratch_size)
    measure_scratch_size = property(css_emit.measure_scratch_size)
    encode_zero = css_emit.encode_zero
    encode_plus = css_emit.encode_plus
    apply_gate = css_emit.apply_gate
    error_correct = css_emit.error_correct
    measure = css_emit.measure
    noisy_measure = css_emit.noisy_measure


def fresh(size, name='ro'):
    prog = Program()
    raw = prog.declare(name, 'BIT', size)
    return prog, MemoryChunk(raw, 0, size)


# ---- MemoryChunk (test/test_quil_classical.py:115-154) -------------------------------------------------------------------

def test_memory_chunk_semantics():
    mem = MemoryReference("test", 0, 20)
    chunk = MemoryChunk(mem, 10, 20)
    assert (chunk.start, chunk.end, len(MemoryChunk(mem, 1, 10))) == (10, 20, 9)
    with pytest.raises(IndexError):
        MemoryChunk(mem, 0, 21)
    assert isinstance(chunk[5], MemoryReference) and chunk[5] == mem[15]
    for sl, want in ((slice(2, 9), (12, 19)), (slice(None, 9), (10, 19)), (slice(2, None), (12, 20))):
        sub = chunk[sl]
        assert isinstance(sub, MemoryChunk) and (sub.start, sub.end) == want
    with pytest.raises(IndexError):
        chunk[10]
    with pytest.raises(IndexError):
        chunk[2:11]
    assert str(chunk) == "test[10:20]" and repr(chunk[1:3]) == "<MChunk test[11:13]>"
    assert [m.offset for m in chunk[7:]] == [17, 18, 19]


# ---- the four generators against their definitions (test/test_quil_classical.py:15-112) -----------------------------------

def test_matmul_is_the_syndrome_product_and_counts_its_instructions():
    rng = np.random.default_rng(3)
    for (m, n) in ((20, 10), (3, 7), (10, 15), (1, 1)):
        mat, vec = rng.integers(0, 2, (m, n)), rng.integers(0, 2, n)
        prog, mem = fresh(n + m + 1)
        vec_in, vec_out, scratch = mem[0:n], mem[n:n + m], mem[n + m:n + m + 1]
        prog += (gates.MOVE(vec_in[i], int(vec[i])) for i in range(n))
        before = len(prog)
        quil_classical.matmul(prog, mat, vec_in, vec_out, scratch)
        assert len(prog) - before == 3 * m * n + m                       # quil_classical.py:74-79
        out = quil_sim.run(prog)['ro']
        assert np.array_equal(out[n:n + m], np.mod(np.matmul(mat, vec), 2))      # test/test_quil_classical.py:38
    prog, mem = fresh(8)
    with pytest.raises(ValueError, match="incompatible"):
        quil_classical.matmul(prog, np.zeros((2, 3), dtype=int), mem[0:2], mem[2:4], mem[4:5])
    with pytest.raises(ValueError, match="incompatible"):
        quil_classical.matmul(prog, np.zeros((2, 3), dtype=int), mem[0:3], mem[3:6], mem[6:7])
    with pytest.raises(ValueError, match="too small"):
        quil_classical.matmul(prog, np.zeros((2, 3), dtype=int), mem[0:3], mem[3:5], mem[5:5])


def test_matmul_emits_the_reference_text():
    prog, mem = fresh(4)
    quil_classical.matmul(prog, np.array([[1, 0]]), mem[0:2], mem[2:3], mem[3:4])
    assert prog.out() == ("DECLARE ro BIT[4]\nMOVE ro[2] 0\nMOVE ro[3] ro[0]\nAND ro[3] 1\nXOR ro[2] ro[3]\n"
                          "MOVE ro[3] ro[1]\nAND ro[3] 0\nXOR ro[2] ro[3]\n")


def test_string_match_conditional_xor_majority_vote():
    cases = [([0] * 8, [0] * 8, True), ([0] * 7 + [1], [0] * 7 + [1], True), ([0] * 6 + [1, 1], [0] * 6 + [1, 1], True),
             ([0] * 8, [0] * 7 + [1], False), ([0] * 6 + [1, 0], [0] * 7 + [1], False), ([0] * 6 + [1, 1], [0] * 7 + [1], False)]
    for vec1, vec2, want in cases:                                       # test/test_quil_classical.py:43-71
        n = len(vec1)
        prog, mem = fresh(n + 2)
        prog += (gates.MOVE(mem[i], vec2[i]) for i in range(n))
        before = len(prog)
        quil_classical.string_match(prog, mem[0:n], np.array(vec1), mem[n:n + 1], mem[n + 1:n + 2])
        assert len(prog) - before == 3 * n + 2
        assert bool(quil_sim.run(prog)['ro'][n]) == want
    rng = np.random.default_rng(0)
    for flag in (0, 1):
        start, vec = rng.integers(0, 2, 9), rng.integers(0, 2, 9)
        prog, mem = fresh(11)
        prog += (gates.MOVE(mem[i], int(start[i])) for i in range(9))
        prog += gates.MOVE(mem[9], flag)
        quil_classical.conditional_xor(prog, mem[0:9], vec, mem[9:10], mem[10:11])
        assert np.array_equal(quil_sim.run(prog)['ro'][:9], start ^ (vec * flag))
    votes = [([0, 0, 0], 0), ([0, 0, 1], 0), ([0, 1, 0], 0), ([1, 0, 0], 0), ([0, 1, 1], 1), ([1, 0, 1], 1), ([1, 1, 0], 1),
             ([1, 1, 1], 1), ([0, 1, 0, 1, 0], 0), ([1, 0, 1, 0, 1], 1)]
    for inputs, want in votes:                                           # test/test_quil_classical.py:73-107
        prog, mem = fresh(len(inputs) + 1)
        raw_int = prog.declare('scratch_int', 'INTEGER', 2)
        prog += (gates.MOVE(mem[1 + i], inputs[i]) for i in range(len(inputs)))
        quil_classical.majority_vote(prog, mem[1:], mem[0], MemoryChunk(raw_int, 0, 2))
        assert quil_sim.run(prog)['ro'][0] == want
    with pytest.raises(ValueError, match="odd"):
        quil_classical.majority_vote(Program(), fresh(4)[1], MemoryReference('x')[0], MemoryChunk(MemoryReference('s', 0, 2), 0, 2))


# ---- the classical decode: emitted code == table lookup ---------------------------------------------------------------------

def decode_with_emitted_code(check, table, noisy_words, known_errors=None):
    """Runs quil_classical_correct on every row of noisy_words; returns the errors registers afterwards."""
    m, n = check.shape
    out = []
    for k, word in enumerate(noisy_words):
        prog, mem = fresh(2 * n + m + 2)
        codeword, errors, scratch = mem[0:n], mem[n:2 * n], mem[2 * n:]
        prog += (gates.MOVE(codeword[i], int(word[i])) for i in range(n))
        if known_errors is not None:
            prog += (gates.MOVE(errors[i], int(known_errors[k][i])) for i in range(n))
        css_emit.quil_classical_correct(prog, codeword, errors, scratch, check, table)
        ro = quil_sim.run(prog)['ro']
        out.append((ro[n:2 * n].copy(), ro[0:n].copy()))
    return out


def test_classical_correct_and_detect_on_the_oracle_tables(steane_h, rm15):
    rng = np.random.default_rng(4)
    for code in (cpu_ref.CSSCode(steane_h, steane_h), cpu_ref.CSSCode(*rm15)):
        for check, table in ((code.parity_check_c2, code._c2_syndromes), (code.parity_check_c1, code._c1_syndromes)):
            m, n = check.shape
            words = [np.zeros(n, dtype=int)] + [np.eye(n, dtype=int)[q] for q in range(n)] + \
                    [(rng.random(n) < 0.2).astype(int) for _ in range(6)]
            for (errs, cw), word in zip(decode_with_emitted_code(check, table, words), words):
                key = int(cpu_ref.vec_to_int(np.mod(check @ word, 2)))
                want = table[key] if key in table else np.zeros(n, dtype=int)       # css_code.py:655-657
                assert np.array_equal(errs, want)
                assert np.array_equal(cw, word ^ want)                               # css_code.py:684-685
            # detect: outcome = [syndrome != 0], with a known error folded in first
            for word in words[:4] + words[-3:]:
                known = (rng.random(n) < 0.1).astype(int)
                prog, mem = fresh(2 * n + m + 3)
                prog += (gates.MOVE(mem[i], int(word[i])) for i in range(n))
                prog += (gates.MOVE(mem[n + i], int(known[i])) for i in range(n))
                css_emit.quil_classical_detect(prog, mem[0:n], mem[n:2 * n], mem[2 * n], mem[2 * n + 1:], check)
                ro = quil_sim.run(prog)['ro']
                assert ro[2 * n] == int(np.any(np.mod(check @ (word ^ known), 2)))
                assert np.array_equal(ro[0:n], word)


def test_classical_correct_instruction_count(steane_h):
    code = cpu_ref.CSSCode(steane_h, steane_h)
    m, n = code.parity_check_c2.shape
    prog, mem = fresh(2 * n + m + 2)
    before = len(prog)
    css_emit.quil_classical_correct(prog, mem[0:n], mem[n:2 * n], mem[2 * n:], code.parity_check_c2, code._c2_syndromes)
    entries = len(code._c2_syndromes)
    assert len(prog) - before == 3 * n + (3 * m * n + m) + entries * ((3 * m + 2) + 3 * n)      # css_code.py:664-685


# ---- CodeBlock, encoders, gates -------------------------------------------------------------------------------------------------

def make_block(prog, n, name):
    raw = prog.declare(name, 'BIT', 2 * n)
    mem = MemoryChunk(raw, 0, 2 * n)
    return CodeBlock(list(range(0, 0)) or [QubitPlaceholder() for _ in range(n)], mem[:n], mem[n:])


def test_code_block_reset_and_size_checks():
    prog = Program()
    block = make_block(prog, 3, 'blk')
    block.reset(prog)
    ops = [inst[0] for inst in prog.instructions]
    assert ops[:4] == ["DECLARE", "MEASURE", "MEASURE", "MEASURE"]
    assert ops.count("JUMP-WHEN") == 3 and ops.count("MOVE") == 6          # qecc.py:44-49
    with pytest.raises(ValueError, match="x_errors"):
        CodeBlock([0, 1], MemoryChunk(MemoryReference('a', 0, 4), 0, 1), MemoryChunk(MemoryReference('a', 0, 4), 1, 3))
    # a flipped qubit reads 1, is flipped back and both registers are cleared
    run_prog = Program(gates.X(1))
    blk = CodeBlock([0, 1, 2], MemoryChunk(run_prog.declare('e', 'BIT', 6), 0, 3), MemoryChunk(MemoryReference('e', 0, 6), 3, 6))
    blk.reset(run_prog)
    ro = run_prog.declare('ro', 'BIT', 3)
    run_prog += (gates.MEASURE(q, ro[q]) for q in range(3))
    mem = quil_sim.run(run_prog)
    assert list(mem['ro']) == [0, 0, 0] and not mem['e'].any()


def test_apply_gate_paulis_transversals_and_refusals(steane_h):
    code = OracleCode(steane_h, steane_h)
    prog = Program()
    a, b = make_block(prog, 7, 'a'), make_block(prog, 7, 'b')
    start = len(prog)
    code.apply_gate(prog, 'X', a)
    assert [(i[1], a.qubits.index(i[2][0])) for i in prog.instructions[start:]] == [('X', 3), ('X', 4), ('X', 6)]   # test_css_code.py:49-53
    start = len(prog)
    code.apply_gate(prog, 'Y', a)
    assert [(i[1], a.qubits.index(i[2][0])) for i in prog.instructions[start:]] == \
        [('X', 3), ('X', 4), ('Y', 6), ('Z', 1), ('Z', 2)]       # test_css_code.py:55-59's factors, in the order of i * X_op * Z_op
    # css_code.py:163-172 asserts the coefficient of i * X_op * Z_op: one shared qubit (mod 4), as here; a code whose logical
    # X and Z share none raises AssertionError in the reference, and here
    shifted = OracleCode(steane_h, steane_h)
    shifted.x_operator_matrix = lambda: np.array([[0, 0, 0, 1, 1, 1, 0]])
    with pytest.raises(AssertionError):
        shifted.apply_gate(Program(), 'Y', a)
    start = len(prog)
    code.apply_gate(prog, 'CNOT', a, b)
    assert [(i[1], i[2]) for i in prog.instructions[start:]] == [('CNOT', (a.qubits[q], b.qubits[q])) for q in range(7)]
    start = len(prog)
    code.apply_gate(prog, 'S', a)                                        # css_code.py:427-431: Z then S on every qubit
    assert [i[1] for i in prog.instructions[start:]] == ['Z', 'S'] * 7
    with pytest.raises(UnsupportedGateError):
        code.apply_gate(prog, 'T', a)
    cols = np.arange(1, 16)
    h1 = np.array([(cols >> k) & 1 for k in range(4)])
    rm = OracleCode(h1, np.vstack([h1] + [h1[i] & h1[j] for i in range(4) for j in range(i + 1, 4)]))
    with pytest.raises(UnsupportedGateError):                            # Reed-Muller: H is not transversal
        rm.apply_gate(Program(), 'H', make_block(Program(), 15, 'x'))


# ---- ftqc.rewrite_program end to end (test/test_ftqc.py) ---------------------------------------------------------------------

def logical(*ops, measure=(0,), bits=1):
    raw = Program()
    ro = raw.declare('ro', 'BIT', bits)
    for op in ops:
        raw += op
    for k, q in enumerate(measure):
        raw += gates.MEASURE(q, ro[k])
    return raw


FTQC_CASES = [
    ("XXX", [gates.X(0)] * 3, 1),                  # test_ftqc.py:25-37
    ("Y", [gates.Y(0)], 1),                        # :39-49
    ("YZ", [gates.Y(0), gates.Z(0)], 1),           # :75-86
    ("HZH", [gates.H(0), gates.Z(0), gates.H(0)], 1),   # :88-100
    ("I", [gates.I(0)], 0),
    ("Z", [gates.Z(0)], 0),
    ("SS on |+> then H", [gates.H(0), gates.S(0), gates.S(0), gates.H(0)], 1),
]


def run_ftqc(code, raw, seed=0, faults=None):
    new_prog = ftqc.rewrite_program(raw, code)
    return new_prog, quil_sim.run(new_prog, seed=seed, faults=faults)


@pytest.mark.parametrize("case", FTQC_CASES, ids=[c[0] for c in FTQC_CASES])
def test_rewritten_programs_measure_the_logical_bit(case, steane_h):
    code = OracleCode(steane_h, steane_h)
    _, ops, want = case
    for seed in range(2):
        new_prog, mem = run_ftqc(code, logical(*ops), seed=seed)
        assert mem['ro'][0] == want
    assert len(new_prog.get_qubits()) == 21                             # one logical qubit + two ancilla blocks of 7


def test_rewritten_program_layout_and_refusals(steane_h):
    code = OracleCode(steane_h, steane_h)
    new_prog = ftqc.rewrite_program(logical(gates.X(0)), code)
    declared = [(i[1], i[2], i[3]) for i in new_prog.instructions if i[0] == "DECLARE"]
    n, t = 7, 1
    assert declared == [("logical_qubit_0", "BIT", 14), ("ancilla_1", "BIT", 14), ("ancilla_2", "BIT", 14),
                        ("scratch", "BIT", 2 * n - 3 + 4 + 2 * t + 1), ("scratch_int", "INTEGER", 2),
                        ("error_correct_scratch", "BIT", 2 * n - 3 + 4), ("ro", "BIT", 1)]      # ftqc.py:54-75, css_code.py:591-597
    assert all(isinstance(q, int) for q in new_prog.get_qubits())

    class TwoLogical(object):
        k = 2
    with pytest.raises(UnsupportedQECCError):
        ftqc.rewrite_program(logical(gates.X(0)), TwoLogical())
    with pytest.raises(UnsupportedGateError):
        ftqc.rewrite_program(logical(("GATE", "T", (0,))), code)


def test_rewritten_program_with_classical_control(steane_h):
    # test_ftqc.py:102-113: H, measure, flip back when 1, measure again -> always 0
    code = OracleCode(steane_h, steane_h)
    raw = Program()
    ro = raw.declare('ro', 'BIT', 2)
    raw += gates.H(0)
    raw += gates.MEASURE(0, ro[0])
    raw.if_then(ro[0], gates.X(0), Program())
    raw += gates.MEASURE(0, ro[1])
    seen = set()
    for seed in range(4):
        _, mem = run_ftqc(code, raw, seed=seed)
        assert mem['ro'][1] == 0
        seen.add(int(mem['ro'][0]))
    assert seen == {0, 1}


def test_two_logical_qubits_superdense_coding(steane_h):
    # test_ftqc.py:115-153, skipped there ("2 qubits is too slow" on a 28-qubit state vector); a stabiliser tableau does not mind
    code = OracleCode(steane_h, steane_h)
    for bit0, bit1 in ((0, 0), (0, 1), (1, 0), (1, 1)):
        ops = [gates.H(0), gates.CNOT(0, 1)] + ([gates.X(0)] if bit1 else []) + ([gates.Z(0)] if bit0 else []) + \
              [gates.CNOT(0, 1), gates.H(0)]
        new_prog, mem = run_ftqc(code, logical(*ops, measure=(0, 1), bits=2), seed=bit0 * 2 + bit1)
        assert (mem['ro'][0], mem['ro'][1]) == (bit0, bit1)
        assert len(new_prog.get_qubits()) == 28


def test_a_single_physical_fault_is_corrected(steane_h):
    # a Pauli error on one data qubit between the logical gate and its round of error correction must not change the answer
    code = OracleCode(steane_h, steane_h)
    raw = logical(gates.X(0))
    new_prog = ftqc.rewrite_program(raw, code)
    insts = new_prog.instructions
    is_x = [i[0] == "GATE" and i[1] == "X" for i in insts]
    first = next(pc for pc in range(len(insts) - 2) if all(is_x[pc:pc + 3]))          # the logical X: X3 X4 X6 in a row
    block = sorted({q for i in insts[:first] if i[0] == "MEASURE" and i[2].name == "logical_qubit_0" for q in [i[1]]})
    assert len(block) == 7 and [insts[first + k][2][0] for k in range(3)] == [block[3], block[4], block[6]]
    for pauli in ("X", "Z", "Y"):
        for victim in (block[0], block[5]):
            mem = quil_sim.run(new_prog, seed=5, faults={first + 3: [(pauli, victim)]})
            assert mem['ro'][0] == 1


# ---- the same through the product's GPU-built CSSCode ---------------------------------------------------------------------------

@pytest.mark.gpu
def test_emitted_decode_equals_the_gpu_syndrome_path(steane_h, rm15):
    from quantum_css_codes_amd import _native
    from quantum_css_codes_amd.css_code import CSSCode, syndrome_batch
    ctx = _native.default_context()
    for code in (CSSCode(steane_h, steane_h), CSSCode(*rm15)):
        n = code.n
        ex = ctx.alloc(64 * 8)
        ez = ctx.alloc(64 * 8)
        ctx.sample_errors_dev(n, 77, 1000, 64, 0.06, 0.03, 0.05, ex, ez, 1)
        for buf, check, table in ((ex, code.parity_check_c2, code._c2_syndromes), (ez, code.parity_check_c1, code._c1_syndromes)):
            words = _native.unpack_rows(buf.download((64, 1), "<u8"), n)
            syn = syndrome_batch(check, words)                              # the GPU product, css_code.py:728
            for (errs, _), word, s in zip(decode_with_emitted_code(check, table, words), words, syn):
                key = int(cpu_ref.vec_to_int(s))
                want = np.asarray(table[key]) if key in table else np.zeros(n, dtype=int)
                assert np.array_equal(errs, want)


@pytest.mark.gpu
def test_ftqc_on_the_gpu_built_code(steane_h):
    from quantum_css_codes_amd.css_code import CSSCode
    code = CSSCode(steane_h, steane_h)
    for _, ops, want in FTQC_CASES[:4]:
        _, mem = run_ftqc(code, logical(*ops), seed=1)
        assert mem['ro'][0] == want
    prog = Program()
    blk = make_block(prog, 7, 'b')
    assert np.array_equal(np.array([(i[1], blk.qubits.index(i[2][0])) for i in code.noisy_encode_zero_program(blk.qubits).instructions[:3]],
                                   dtype=object), np.array([('H', 0), ('H', 1), ('H', 2)], dtype=object))
