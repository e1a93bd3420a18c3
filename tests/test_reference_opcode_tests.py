"""The reference's opcode unit tests, one to one ("Opcode unit tests on hand-built ACIR"):
plonky2-backend/src/circuit_translation/tests/test_assert_zero.rs (10 tests + the field pin) and tests/test_blackbox.rs
(RANGE 8 / 16 / 32 / 33 bits with their `should_panic` negatives, the 64-bit refusal, AND / XOR at 8 / 16 / 32 bits), with
the circuits of tests/factories/circuit_factory.rs restated as data (coefficients, witness indices, public parameters) and the
witness values of each test.  Every reference test is "build circuit -> real prove -> real verify" with
`proof.public_inputs[i] == expected`; so is every test here: restated translator (translate.py, public parameters hashed in
circuit like build() does) -> p2gpu_build_blob -> prove -> verify, on the CPU oracle, and all of them again on the MI355X with
the oracle's bytes (`-m gpu`)."""
import numpy as np
import pytest

from conftest import P


def _pi_tail(pis):
    return b"".join(int(v).to_bytes(8, "little") for v in pis)


# (test name in the reference, opcodes, public parameters, witness assignment, expected public inputs)
def _assert_zero_cases():
    w = list(range(4))
    quad = [(2, 0, 0), (3, 0, 1), (4, 1, 2), (5, 2, 3), (6, 3, 3), (7, 1, 1)]
    return [
        ("assert_x_equals_zero", [("assert_zero", [], [(1, 0)], 0)], [0], {0: 0}, [0]),                                   # :6-25
        ("assert_x_equals_constant", [("assert_zero", [], [(1, 0)], P - 4)], [0], {0: 4}, [4]),                           # :28-47
        ("assert_c_times_x_equals_constant", [("assert_zero", [], [(3, 0)], P - 12)], [0], {0: 4}, [4]),                  # :50-69
        ("x_times_3_plus_y_times_4_equals_constant", [("assert_zero", [], [(3, 0), (9, 1)], P - 12)], [0, 1], {0: 1, 1: 1}, [1, 1]),   # :72-103
        ("multiple_linear_combinations", [("assert_zero", [], [(3, i) for i in reversed(w)], P - 12)], w, {i: 1 for i in w}, [1] * 4),  # :106-134
        ("x_times_x_equals_constant", [("assert_zero", [(2, 0, 0)], [], P - 0x20)], [0], {0: 4}, [4]),                    # :137-157
        ("c_times_x_times_y_equals_constant", [("assert_zero", [(2, 0, 1)], [], P - 0x28)], [0, 1], {0: 4, 1: 5}, [4, 5]),  # :160-183
        ("multiple_cuadratic_terms", [("assert_zero", quad, [], P - 0x6c)], w, {i: 2 for i in w}, [2] * 4),               # :186-214
        ("multiple_cuadratic_terms_and_linear_combinations",
         [("assert_zero", quad, [(1, 0), (2, 1), (3, 2), (4, 3)], P - 0x80)], w, {i: 2 for i in w}, [2] * 4),             # :217-245
        ("circuits_with_2_assert_zero_opcodes",
         [("assert_zero", [], [(1, 0), (P - 1, 1)], 4), ("assert_zero", [(1, 1, 1)], [], P - 0x19)], [0], {0: 1, 1: 5}, [1]),  # :248-272
    ]


def _bitwise_cases():
    out = []
    for op, bits, a, b, r in (("and", 8, 5, 3, 1), ("and", 16, 0xFF00, 0xF0F0, 0xF000), ("and", 32, 0xFF00FF00, 0xF0F0F0F0, 0xF000F000),
                              ("xor", 8, 3, 5, 6), ("xor", 16, 0xFF00, 0xF0F0, 0x0FF0), ("xor", 32, 0xFF00FF00, 0xF0F0F0F0, 0x0FF00FF0)):
        # circuit_factory.rs bitwise_{and,xor}_circuit: RANGE on both inputs, then the operation into witness 2 (a return value: private)
        ops = [("range", 0, bits), ("range", 1, bits), (op, 0, 1, 2, bits)]
        out.append((f"bitwise_{op}_up_to_{bits}_bits", ops, [0, 1], {0: a, 1: b, 2: r}, [a, b]))   # test_blackbox.rs:111-217
    return out


def _range_cases():
    return [(f"range_check_u{bits}", [("range", 0, bits)], [0], {0: (1 << bits) - 1}, [(1 << bits) - 1]) for bits in (8, 16, 32, 33)]  # :8-75


CASES = _assert_zero_cases() + _range_cases() + _bitwise_cases()


def _build(pkg, ops, public, witness):
    cb = pkg.translate.CircuitBuilderFromAcirToPlonky2()
    cb.translate_circuit(ops, public_parameters=public)
    blob, wires = cb.build(witness)
    return cb, blob, wires


@pytest.mark.parametrize("name,ops,public,witness,expected", CASES, ids=[c[0] for c in CASES])
def test_reference_opcode_test_on_the_oracle(pkg, orc, name, ops, public, witness, expected):
    cb, blob, wires = _build(pkg, ops, public, witness)
    pis = cb.public_inputs()
    assert pis == expected                                 # assert_eq!(expected, proof.public_inputs[i])
    oc = orc.OracleCircuit(blob)
    proof, _ = oc.prove(wires, public_inputs=pis)
    assert proof.endswith(_pi_tail(expected)) and oc.verify(proof)   # assert!(circuit_data.verify(proof).is_ok())
    oc.close()


@pytest.mark.parametrize("bits", [8, 16, 32])
def test_witness_bigger_than_the_range_is_refused(pkg, bits):
    """test_blackbox.rs:17-24, 36-43, 55-62 (#[should_panic]): 2^bits does not fit a bits-wide range check -- upstream's witness
    generator panics, the restated builder raises."""
    with pytest.raises(ValueError):
        _build(pkg, [("range", 0, bits)], [0], {0: 1 << bits})


def test_range_checks_of_64_bits_are_not_supported(pkg):
    """test_blackbox.rs:75-83: the reference refuses RANGE above 33 bits with this message (circuit_translation/mod.rs:131-139)."""
    with pytest.raises(AssertionError, match="Range checks with more than 33 bits are not allowed yet while using Plonky2 prover"):
        _build(pkg, [("range", 0, 64)], [0], {0: (1 << 64) - (1 << 32)})


def test_current_noir_is_using_goldilocks_field():
    """test_assert_zero.rs:275-285: the field modulus pin."""
    assert P == 18446744069414584321 == 2 ** 64 - 2 ** 32 + 1


def test_a_wrong_witness_does_not_verify(pkg, orc):
    """What every positive test above implies and none of the reference's states: a witness that does not satisfy the opcode
    yields no acceptable proof (3 x = 12 with x = 5)."""
    with pytest.raises(ValueError):
        _build(pkg, [("assert_zero", [], [(3, 0)], P - 12)], [0], {0: 5})


@pytest.mark.gpu
def test_reference_opcode_tests_on_the_gpu(pkg, orc):
    """All twenty circuits above through the C ABI on the MI355X: the oracle's bytes, the product's verifier accepts."""
    for name, ops, public, witness, expected in CASES:
        cb, blob, wires = _build(pkg, ops, public, witness)
        pis = cb.public_inputs()
        cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
        want, _ = oc.prove(wires, public_inputs=pis)
        got = cd.prove(wires, public_inputs=pis).to_bytes()
        assert got == want, name
        assert got.endswith(_pi_tail(expected))
        cd.verify(got)
        cd.close()
        oc.close()
