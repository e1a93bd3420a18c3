"""N4: the ACIR program / witness readers (acvm-backend-plonky2_amd/acir.py; the reference's
noir_and_plonky2_serialization.rs:42-64) and the whole front of the `prove` action on them
(actions/prove_action.rs:21-56, 100-116): program JSON + witness file -> opcodes -> circuit -> proof.

No compiled program exists in the reference tree, so the files are made here with the module's own inverse
functions (UNPINNED layout, see its docstring); the programs are the ones the reference's test factories build in
code (circuit_translation/tests/factories/circuit_factory.rs) and the fibonacci example's single opcode."""
import base64
import gzip
import json

import numpy as np
import pytest

from conftest import P


def expr(mul=(), lin=(), q_c=0):
    return {"mul_terms": [tuple(t) for t in mul], "linear_combinations": [tuple(t) for t in lin], "q_c": q_c % P}


def programs():
    """name -> (circuit, witness map)"""
    out = {}
    # circuit_factory.rs:11-40  x == 4 with x public  (here private: public parameters need PoseidonGate rows)
    out["x_equals_4"] = ({"opcodes": [("AssertZero", expr(lin=[(1, 0)], q_c=-4))], "private_parameters": [0]}, {0: 4})
    # example_programs/fibonacci/src/main.nr:1-10 compiles to EXPR [(1, _0) -377]
    out["fibonacci"] = ({"opcodes": [("AssertZero", expr(lin=[(1, 0)], q_c=-377))], "private_parameters": [0],
                         "assert_messages": [(("Acir", 0), ("StaticString", "attempt to add with overflow"))]}, {0: 377})
    # circuit_factory.rs:196-236: x * x == 4 * y style: one quadratic and one linear opcode, an intermediate witness
    out["quadratic"] = ({"current_witness_index": 2,
                         "opcodes": [("AssertZero", expr(mul=[(1, 0, 1)], lin=[(P - 1, 2)])),
                                     ("AssertZero", expr(lin=[(3, 0), (2, 1)], q_c=-17))],
                         "private_parameters": [0, 1]}, {0: 3, 1: 4, 2: 12})
    # circuit_factory.rs:237-330: two RANGEs and an AND / XOR, plus what nargo puts around them: a Brillig call that
    # computes the output off-circuit, a ToLeRadix directive and a Call -- all ignored by the backend (mod.rs:98-104)
    fi = lambda w, bits: (w, bits)  # noqa: E731
    a, c = 0xB7, 0x5D
    bitwise = {"current_witness_index": 5,
               "opcodes": [("BrilligCall", {"id": 0, "inputs": [("Single", expr(lin=[(1, 0)])), ("Array", [expr(lin=[(1, 2)])]), ("MemoryArray", 7)],
                                            "outputs": [("Simple", 3), ("Array", [4, 5])], "predicate": expr(q_c=1)}),
                           ("BlackBoxFuncCall", {"name": "RANGE", "input": fi(0, 8)}),
                           ("BlackBoxFuncCall", {"name": "RANGE", "input": fi(2, 8)}),
                           ("Directive", {"name": "ToLeRadix", "a": expr(lin=[(1, 0)]), "b": [6, 7], "radix": 256}),
                           ("BlackBoxFuncCall", {"name": "AND", "lhs": fi(0, 8), "rhs": fi(2, 8), "output": 3}),
                           ("BlackBoxFuncCall", {"name": "XOR", "lhs": fi(0, 8), "rhs": fi(2, 8), "output": 4}),
                           ("AssertZero", expr(lin=[(1, 3), (1, 4), (P - 1, 5)]))],
               "expression_width": ("Bounded", 4), "private_parameters": [0, 2], "return_values": [5]}
    out["bitwise"] = (bitwise, {0: a, 2: c, 3: a & c, 4: a ^ c, 5: (a & c) + (a ^ c)})
    return out


def test_program_and_witness_files_round_trip(pkg, tmp_path):
    ac = pkg.acir
    for name, (circuit, witness) in programs().items():
        path = tmp_path / (name + ".json")
        path.write_text(ac.program_json([circuit]))
        prog = ac.deserialize_program_within_file_path(str(path))
        assert len(prog["functions"]) == 1
        got = prog["functions"][0]
        assert got["opcodes"] == circuit["opcodes"], name
        assert got["current_witness_index"] == circuit.get("current_witness_index", 0)
        assert got["private_parameters"] == sorted(circuit.get("private_parameters", []))
        assert got["return_values"] == sorted(circuit.get("return_values", [])) and got["public_parameters"] == []
        assert got["expression_width"] == tuple(circuit.get("expression_width", ("Unbounded",)))
        assert got["assert_messages"] == circuit.get("assert_messages", []) and got["recursive"] is False
        wpath = tmp_path / (name + ".gz")
        wpath.write_bytes(ac.serialize_witness_stack([{"index": 0, "witness": witness}]))
        stack = ac.deserialize_witnesses_within_file_path(str(wpath))
        assert stack == [{"index": 0, "witness": witness}]
    # layout fixed points of bincode 1.3.3: u64 lengths, u32 tags, field elements as hex strings
    raw = gzip.decompress(ac.serialize_program([programs()["x_equals_4"][0]]))
    assert raw[:8] == (1).to_bytes(8, "little") and raw[8:12] == bytes(4) and raw[12:20] == (1).to_bytes(8, "little")
    assert raw[20:24] == bytes(4)                                   # Opcode::AssertZero
    assert raw[24:32] == bytes(8) and raw[32:40] == (1).to_bytes(8, "little")   # no mul terms, one linear term
    assert raw[40:48] == (64).to_bytes(8, "little") and raw[48:112] == b"0" * 63 + b"1"   # FieldElement::one().to_hex()
    # a field element wider than Goldilocks is reduced like the backend does (assert_zero_translator.rs:118-121)
    big = (1 << 200) + 12345
    doc = bytearray(raw)
    doc[48:112] = ("%064x" % big).encode()
    got = ac.deserialize_program(gzip.compress(bytes(doc)))["functions"][0]["opcodes"][0][1]["linear_combinations"][0][0]
    assert got == big % P


def test_malformed_files_are_refused_not_crashed_on(pkg, tmp_path):
    ac = pkg.acir
    circuit, witness = programs()["bitwise"]
    good = gzip.decompress(ac.serialize_program([circuit]))
    with pytest.raises(ac.AcirFormatError):
        ac.deserialize_program(b"not gzip")
    (tmp_path / "a.json").write_text(json.dumps({"bytecode": "@@@"}))
    with pytest.raises(ac.AcirFormatError):
        ac.deserialize_program_within_file_path(str(tmp_path / "a.json"))
    (tmp_path / "b.json").write_text(json.dumps({"abi": {}}))
    with pytest.raises(ac.AcirFormatError, match="Expected a different circuit format"):
        ac.deserialize_program_within_file_path(str(tmp_path / "b.json"))
    rng = np.random.default_rng(3)
    refused = parsed = 0
    for it in range(1500):
        m = bytearray(good)
        if it % 3 == 0:
            m = m[:int(rng.integers(0, len(m)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        try:
            ac.deserialize_program(gzip.compress(bytes(m)))
            parsed += 1
        except ac.AcirFormatError:
            refused += 1
    assert refused > 300 and parsed + refused == 1500
    wgood = gzip.decompress(ac.serialize_witness_stack([{"index": 0, "witness": witness}]))
    for cut in (0, 5, 9, len(wgood) - 1):
        with pytest.raises(ac.AcirFormatError):
            ac.deserialize_witnesses(gzip.compress(wgood[:cut]))
    with pytest.raises(ac.AcirFormatError):
        ac.deserialize_witnesses(gzip.compress(wgood + b"\0"))


def _front_of_prove_action(pkg, tmp_path, name):
    """prove_action.rs:21-56: read the program, translate functions[0], build; :100-116: pop the witness stack and
    assign every witness to its target."""
    ac, tr = pkg.acir, pkg.translate
    circuit, witness = programs()[name]
    (tmp_path / "p.json").write_text(ac.program_json([circuit]))
    (tmp_path / "w.gz").write_bytes(ac.serialize_witness_stack([{"index": 0, "witness": witness}]))
    prog = ac.deserialize_program_within_file_path(str(tmp_path / "p.json"))
    stack = ac.deserialize_witnesses_within_file_path(str(tmp_path / "w.gz"))
    cb = tr.CircuitBuilderFromAcirToPlonky2()
    cb.translate_circuit(ac.to_translator_opcodes(prog["functions"][0]), prog["functions"][0]["public_parameters"])
    return cb, cb.build(stack.pop()["witness"])


@pytest.mark.parametrize("name", ["fibonacci", "quadratic", "bitwise"])
def test_files_to_proof_on_the_oracle(pkg, orc, tmp_path, name):
    cb, (blob, wires) = _front_of_prove_action(pkg, tmp_path, name)
    oc = orc.OracleCircuit(blob)
    proof, _ = oc.prove(wires)
    assert oc.verify(proof)
    if name == "fibonacci":
        import sys
        from conftest import GOLDEN
        sys.path.insert(0, GOLDEN)
        import mini_builder
        assert blob.tobytes() == mini_builder.fibonacci()[0].tobytes()   # the same circuit as the hand-written one
    # a witness file that does not satisfy the program is refused when the wires are filled
    ac = pkg.acir
    circuit, witness = programs()[name]
    bad = dict(witness)
    bad[0] = (bad[0] + 1) % P
    cb2 = pkg.translate.CircuitBuilderFromAcirToPlonky2()
    cb2.translate_circuit(ac.to_translator_opcodes(circuit))
    with pytest.raises(ValueError):
        cb2.build(bad)


def test_unsupported_opcodes_are_named(pkg):
    ac = pkg.acir
    circuit = {"opcodes": [("MemoryInit", {"block_id": 0, "init": [0, 1], "block_type": "Memory"})]}
    prog = ac.deserialize_program(ac.serialize_program([circuit]))
    with pytest.raises(NotImplementedError, match="MemoryInit"):
        ac.to_translator_opcodes(prog["functions"][0])
    circuit = {"opcodes": [("BlackBoxFuncCall", {"name": "SHA256", "inputs": [(0, 8)], "outputs": list(range(1, 33))})]}
    prog = ac.deserialize_program(ac.serialize_program([circuit]))
    with pytest.raises(NotImplementedError, match="SHA256"):
        ac.to_translator_opcodes(prog["functions"][0])


@pytest.mark.gpu
def test_files_to_proof_on_the_gpu(pkg, orc, tmp_path):
    """The same front end, proved through the C ABI on the MI355X: bytes identical to the oracle's."""
    for name in ("fibonacci", "bitwise"):
        cb, (blob, wires) = _front_of_prove_action(pkg, tmp_path, name)
        expect, _ = orc.OracleCircuit(blob).prove(wires)
        cd = pkg.CircuitData(blob)
        proof = cd.prove(wires)
        assert proof.to_bytes() == expect
        cd.verify(proof)
        cd.close()
    assert base64.b64decode(json.loads(pkg.acir.program_json([programs()["fibonacci"][0]]))["bytecode"])[:2] == b"\x1f\x8b"
