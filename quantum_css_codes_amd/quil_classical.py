"""
Classical-memory code generators: the in-circuit statement of the syndrome product and of the table
decode, pyQuil-free.  Mirrors quil_classical.py of the reference -- same names, argument meaning, error behaviour and, above
all, the same instruction stream (quil_classical.py:60-127), built on quantum_css_codes_amd.quil instead of pyQuil.

What the emitted code computes is exactly what the GPU path computes in bulk:
    matmul            result[i] = XOR_j vec[j] AND mat[i][j]          = np.mod(np.matmul(mat, vec), 2), css_code.py:728
    string_match      output = (mem == vec)                            = the key comparison of the syndrome-table lookup
    conditional_xor   mem ^= flag * vec                                = applying the table's correction
tests/test_quil_emission.py runs the emitted programs (oracle/quil_sim.py) against gf2_syndrome_batch and the device-built
syndrome tables, and checks the instruction counts the reference's loops imply (3 m n + m for matmul).
"""
from .quil import MemoryReference, Program, gates


class MemoryChunk(object):
    """A slice [start, end) of one declared classical register (quil_classical.py:10-57): sub-slicing, indexing into
    MemoryReferences, iteration; out-of-range access raises IndexError."""

    def __init__(self, mem, start, end):
        if mem.declared_size is not None and mem.declared_size < end:
            raise IndexError("bounds would exceed declared size of memory reference")
        self.mem, self.start, self.end = mem, start, end

    def _span(self):
        return "%s[%d:%d]" % (self.mem.name, self.start + self.mem.offset, self.end + self.mem.offset)

    def __str__(self):
        return self._span()

    def __repr__(self):
        return "<MChunk %s>" % self._span()

    def __len__(self):
        return self.end - self.start

    def __getitem__(self, index):
        if isinstance(index, slice):
            lo = self.start + (0 if index.start is None else index.start)
            hi = self.start + (len(self) if index.stop is None else index.stop)
            if lo < self.start or hi > self.end:
                raise IndexError("out of bounds")
            return MemoryChunk(self.mem, lo, hi)
        if not 0 <= index < len(self):
            raise IndexError("out of bounds")
        return self.mem[self.start + index]

    def __iter__(self):
        return (self.mem[i] for i in range(self.start, self.end))


def matmul(prog, mat, vec, result, scratch):
    """Instructions for result = mat . vec over GF(2), mat a constant m x n binary matrix (quil_classical.py:60-79):
    per row one MOVE result[i] 0, per entry MOVE / AND / XOR through scratch[0] -- 3 m n + m instructions."""
    m, n = mat.shape
    if len(vec) != n:
        raise ValueError("mat and vec are of incompatible sizes")
    if len(result) != m:
        raise ValueError("mat and result are of incompatible sizes")
    if len(scratch) < 1:
        raise ValueError("scratch buffer is too small")
    tmp = scratch[0]
    for i in range(m):
        acc = result[i]
        prog += gates.MOVE(acc, 0)
        for j in range(n):
            prog += (gates.MOVE(tmp, vec[j]), gates.AND(tmp, int(mat[i][j])), gates.XOR(acc, tmp))


def string_match(prog, mem, vec, output, scratch):
    """output[0] = 1 iff the bits of mem equal the constant vec (quil_classical.py:81-97): OR of the XORs, negated."""
    n = len(mem)
    if vec.size != n:
        raise ValueError("length of mem and vec do not match")
    if len(scratch) < 1:
        raise ValueError("scratch buffer is too small")
    tmp, out = scratch[0], output[0]
    prog += gates.MOVE(out, 0)
    for i in range(n):
        prog += (gates.MOVE(tmp, mem[i]), gates.XOR(tmp, int(vec[i])), gates.IOR(out, tmp))
    prog += gates.NOT(out)


def conditional_xor(prog, mem, vec, flag, scratch):
    """mem ^= vec when flag[0] is set, untouched otherwise (quil_classical.py:99-111)."""
    n = len(mem)
    if vec.size != n:
        raise ValueError("length of mem and vec do not match")
    tmp = scratch[0]
    for i in range(n):
        prog += (gates.MOVE(tmp, flag[0]), gates.AND(tmp, int(vec[i])), gates.XOR(mem[i], tmp))


def majority_vote(prog, inputs, output, scratch_int):
    """output = 1 iff more than half of the (odd number of) input bits are set (quil_classical.py:113-127); counts in
    the INTEGER register scratch_int[0], compares with GE."""
    if len(scratch_int) < 2:
        raise ValueError("scratch_int buffer too small")
    if len(inputs) % 2 == 0:
        raise ValueError("inputs length must be odd")
    total, term = scratch_int[0], scratch_int[1]
    prog += gates.MOVE(total, 0)
    for bit in inputs:
        prog += (gates.CONVERT(term, bit), gates.ADD(total, term))
    prog += gates.MOVE(term, (len(inputs) + 1) // 2)
    prog += gates.GE(output, total, term)


__all__ = ["MemoryChunk", "MemoryReference", "Program", "matmul", "string_match", "conditional_xor", "majority_vote"]
