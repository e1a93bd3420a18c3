"""The reference's gadget-internals tests, one to one ("Gadget internals"):
plonky2-backend/src/circuit_translation/tests/test_sha256_internal.rs -- rotate_right / shift_right / choose / majority /
add_module_32_bits of binary_digits_target.rs, each on hand-picked bit vectors, with the `#[should_panic]` negatives.  The
reference builds a CircuitBuilder on standard_recursion_config, applies the gadget to virtual bool targets, assigns inputs AND
claimed outputs and requires prove + verify (harnesses at :307-478).  Same here on the restated builder (translate.py, 135
wires) and the CPU oracle; a wrong claimed output makes upstream's witness generation panic and the restated one raise.
(The full compression KAT of :480-549 is tests/test_translate.py::test_sha256_compression_circuit_reference_vector.)"""
import pytest


def bits(s):
    return [int(c) for c in s]


Z32, O32 = "0" * 32, "1" * 32
ROT = [  # (reference test, n, input, claimed output, ok)
    ("rotate_right_4_1", 1, "0010", "0001", True), ("rotate_right_failed", 1, "0010", "0000", False),               # :12-28
    ("rotate_right_32_1", 1, "0010" * 8, "0001" * 8, True), ("rotate_right_32_2", 2, "1" + "0" * 31, "001" + "0" * 29, True),   # :30-62
    ("rotate_right_32_32", 32, "1" + "0" * 31, "1" + "0" * 31, True)]                                               # :64-78
SHR = [("shift_right_4_1", 1, "1111", "0111", True), ("shift_right_failed", 1, "1111", "0110", False),              # :81-98
       ("shift_right_32_16", 16, "1" * 16 + "0" * 16, "0" * 16 + "1" * 16, True)]                                   # :100-114
CHOOSE = [("choose_4", "0101", "1100", "0011", "0110", True), ("choose_4_failed", "0101", "1100", "0011", "0111", False),   # :117-138
          ("choose_32", Z32, Z32, "01000001000101000010001000000100", "01000001000101000010001000000100", True)]           # :140-160
MAJ = [("majority_4", "0101", "1100", "0011", "0101", True), ("majority_4_failed", "0101", "1100", "0011", "0100", False),  # :163-184
       ("majority_32", Z32, Z32, O32, Z32, True)]                                                                           # :186-202
ADD = [("add_module_32_bits_without_any_carry", Z32, O32, O32, True),                                                # :205-220
       ("add_module_32_bits_fail", "0" * 31 + "1", "1" * 31 + "0", "1" * 31 + "0", False),                           # :223-239
       ("simple_add_module_32_bits_with_carry", "0" * 31 + "1", "0" * 31 + "1", "0" * 30 + "10", True),             # :242-261
       ("flooded_add_module_32_bits_with_carry", "0" * 31 + "1", "0" + "1" * 31, "1" + "0" * 31, True),             # :264-283
       ("add_module_32_bits_with_overflow", "0" * 31 + "1", O32, Z32, True)]                                         # :286-305


def _run(pkg, orc, apply, inputs, claimed, ok):
    tr = pkg.translate
    b = tr.CircuitBuilder(num_wires=135)
    ins = [tr.BinaryDigitsTarget([b.add_virtual_target() for _ in v]) for v in inputs]   # add_virtual_bool_target_unsafe
    out = apply(tr.BinaryDigitsTarget, b, ins)
    wit = {t: v for x, vals in zip(ins, inputs) for t, v in zip(x.bits, vals)}
    for t, v in zip(out.bits, claimed):     # partial_witnesses.set_target(result.bits[i].target, output_values[i])
        if t in wit and wit[t] != v:
            assert not ok
            return
        wit[t] = v
    if not ok:
        with pytest.raises(ValueError):
            b.build(wit)
        return
    blob, wires = b.build(wit)
    oc = orc.OracleCircuit(blob)
    proof, _ = oc.prove(wires)
    assert oc.verify(proof)
    oc.close()


@pytest.mark.parametrize("name,n,x,y,ok", ROT, ids=[c[0] for c in ROT])
def test_rotate_right(pkg, orc, name, n, x, y, ok):
    _run(pkg, orc, lambda B, b, ins: B.rotate_right(ins[0], n % len(x), b), [bits(x)], bits(y), ok)


@pytest.mark.parametrize("name,n,x,y,ok", SHR, ids=[c[0] for c in SHR])
def test_shift_right(pkg, orc, name, n, x, y, ok):
    _run(pkg, orc, lambda B, b, ins: B.shift_right(ins[0], n, b), [bits(x)], bits(y), ok)


@pytest.mark.parametrize("name,c,x,y,z,ok", CHOOSE, ids=[c[0] for c in CHOOSE])
def test_choose(pkg, orc, name, c, x, y, z, ok):
    _run(pkg, orc, lambda B, b, ins: B.choose(ins[0], ins[1], ins[2], b), [bits(c), bits(x), bits(y)], bits(z), ok)


@pytest.mark.parametrize("name,x0,x1,x2,z,ok", MAJ, ids=[c[0] for c in MAJ])
def test_majority(pkg, orc, name, x0, x1, x2, z, ok):
    _run(pkg, orc, lambda B, b, ins: B.majority(ins[0], ins[1], ins[2], b), [bits(x0), bits(x1), bits(x2)], bits(z), ok)


@pytest.mark.parametrize("name,x,y,z,ok", ADD, ids=[c[0] for c in ADD])
def test_add_module_32_bits(pkg, orc, name, x, y, z, ok):
    _run(pkg, orc, lambda B, b, ins: B.add_module_32_bits(ins[0], ins[1], b), [bits(x), bits(y)], bits(z), ok)
