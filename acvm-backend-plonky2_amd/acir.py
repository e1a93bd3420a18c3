"""ACIR program and witness files -> Python data: the readers of the reference's
``noir_and_plonky2_serialization.rs``.

  * ``deserialize_program_within_file_path`` (:42-58): the ``nargo compile`` artefact is JSON whose
    ``"bytecode"`` is base64 of ``Program::serialize_program`` = gzip(bincode(Program));
  * ``deserialize_witnesses_within_file_path`` (:60-64): ``WitnessStack::try_from(bytes)`` =
    gzip(bincode(WitnessStack)), the file ``nargo execute`` writes.

The two crates that own the formats are not in the reference tree: ``acir`` / ``acir_field`` 0.47.0
(plonky2-backend/Cargo.toml:12-13,30-31, a *path* dependency on a noir fork) through ``bincode`` 1.3.3's
default configuration (Cargo.lock:261: little endian, fixed-width integers, u64 lengths, u32 enum tags).
**UNPINNED**: the type definitions below are a recollection of acir 0.47.0
(acvm-repo/acir/src/circuit/{mod,opcodes,black_box_functions,brillig,directives}.rs,
native_types/{expression,witness,witness_stack}.rs); no compiled program or witness file exists under
/root/reference to test them on.  What the reference tree does pin: the field names and shapes its own code and
test factories touch -- ``Circuit { current_witness_index, expression_width, opcodes, private_parameters,
public_parameters, return_values, assert_messages, recursive }`` (circuit_translation/tests/factories/
circuit_factory.rs:17-28), ``Expression { mul_terms, linear_combinations, q_c }`` (:31-40), ``FunctionInput {
witness, num_bits }`` (:191-194), the opcode variants and their fields (circuit_translation/mod.rs:88-190),
``BlockType::Memory`` (tests/test_memory_operations.rs:4), ``witness_stack.pop().unwrap().witness``
(actions/prove_action.rs:108-116) and the conversion of a field element to Goldilocks (big-endian bytes reduced
modulo p, assert_zero_translator.rs:118-121).

The serialisers at the bottom are the inverse functions (what nargo does, not the backend); the tests use them
to make files, in place of the compiled artefacts the tree lacks.

Host-side Python by design, like ``translate.py``: runs once per program, off the hot path.
"""
import base64
import gzip
import json
import struct

P = 0xFFFFFFFF00000001

# BlackBoxFuncCall variants of acir 0.47.0, in declaration order (= bincode tag)
BLACK_BOX = ["AES128Encrypt", "AND", "XOR", "RANGE", "SHA256", "Blake2s", "Blake3", "SchnorrVerify", "PedersenCommitment",
             "PedersenHash", "EcdsaSecp256k1", "EcdsaSecp256r1", "MultiScalarMul", "EmbeddedCurveAdd", "Keccak256",
             "Keccakf1600", "RecursiveAggregation", "BigIntAdd", "BigIntSub", "BigIntMul", "BigIntDiv", "BigIntFromLeBytes",
             "BigIntToLeBytes", "Poseidon2Permutation", "Sha256Compression"]
OPCODES = ["AssertZero", "BlackBoxFuncCall", "Directive", "MemoryOp", "MemoryInit", "BrilligCall", "Call"]


class AcirFormatError(ValueError):
    pass


class _In:
    def __init__(self, data):
        self.d, self.at = data, 0

    def take(self, n):
        if n < 0 or self.at + n > len(self.d):
            raise AcirFormatError("truncated at byte %d" % self.at)
        b = self.d[self.at:self.at + n]
        self.at += n
        return b

    def u8(self):
        return self.take(1)[0]

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def length(self):
        n = self.u64()
        if n > len(self.d) - self.at:  # every element takes at least one byte
            raise AcirFormatError("length %d at byte %d exceeds the input" % (n, self.at - 8))
        return n

    def boolean(self):
        b = self.u8()
        if b > 1:
            raise AcirFormatError("bool byte %d" % b)
        return bool(b)

    def tag(self, names, what):
        t = self.u32()
        if t >= len(names):
            raise AcirFormatError("%s tag %d at byte %d" % (what, t, self.at - 4))
        return names[t]

    def string(self):
        try:
            return self.take(self.length()).decode("utf-8")
        except UnicodeDecodeError as e:
            raise AcirFormatError(str(e))

    def vec(self, item):
        return [item() for _ in range(self.length())]

    def option(self, item):
        t = self.u8()
        if t > 1:
            raise AcirFormatError("Option tag %d" % t)
        return item() if t else None

    # FieldElement: serde writes `to_hex()`; the backend reduces the big-endian value modulo Goldilocks
    def field(self):
        s = self.string()
        try:
            return int(s, 16) % P
        except ValueError:
            raise AcirFormatError("field element %r" % s[:40])

    def witness(self):
        return self.u32()

    def expression(self):
        mul = self.vec(lambda: (self.field(), self.witness(), self.witness()))
        lin = self.vec(lambda: (self.field(), self.witness()))
        return {"mul_terms": mul, "linear_combinations": lin, "q_c": self.field()}

    def function_input(self):
        return (self.witness(), self.u32())  # (witness, num_bits)

    def inputs(self, n=None):
        return [self.function_input() for _ in range(self.length() if n is None else n)]

    def witnesses(self, n=None):
        return [self.witness() for _ in range(self.length() if n is None else n)]


def _black_box(r):
    name = r.tag(BLACK_BOX, "BlackBoxFuncCall")
    f = {"name": name}
    if name == "AES128Encrypt":
        f.update(inputs=r.inputs(), iv=r.inputs(16), key=r.inputs(16), outputs=r.witnesses())
    elif name in ("AND", "XOR"):
        f.update(lhs=r.function_input(), rhs=r.function_input(), output=r.witness())
    elif name == "RANGE":
        f.update(input=r.function_input())
    elif name in ("SHA256", "Blake2s", "Blake3"):
        f.update(inputs=r.inputs(), outputs=r.witnesses(32))
    elif name == "SchnorrVerify":
        f.update(public_key_x=r.function_input(), public_key_y=r.function_input(), signature=r.inputs(64), message=r.inputs(),
                 output=r.witness())
    elif name == "PedersenCommitment":
        f.update(inputs=r.inputs(), domain_separator=r.u32(), outputs=(r.witness(), r.witness()))
    elif name == "PedersenHash":
        f.update(inputs=r.inputs(), domain_separator=r.u32(), output=r.witness())
    elif name in ("EcdsaSecp256k1", "EcdsaSecp256r1"):
        f.update(public_key_x=r.inputs(32), public_key_y=r.inputs(32), signature=r.inputs(64), hashed_message=r.inputs(32),
                 output=r.witness())
    elif name == "MultiScalarMul":
        f.update(points=r.inputs(), scalars=r.inputs(), outputs=(r.witness(), r.witness(), r.witness()))
    elif name == "EmbeddedCurveAdd":
        f.update(input1=r.inputs(3), input2=r.inputs(3), outputs=(r.witness(), r.witness(), r.witness()))
    elif name == "Keccak256":
        f.update(inputs=r.inputs(), var_message_size=r.function_input(), outputs=r.witnesses(32))
    elif name == "Keccakf1600":
        f.update(inputs=r.inputs(25), outputs=r.witnesses(25))
    elif name == "RecursiveAggregation":
        f.update(verification_key=r.inputs(), proof=r.inputs(), public_inputs=r.inputs(), key_hash=r.function_input())
    elif name in ("BigIntAdd", "BigIntSub", "BigIntMul", "BigIntDiv"):
        f.update(lhs=r.u32(), rhs=r.u32(), output=r.u32())
    elif name == "BigIntFromLeBytes":
        f.update(inputs=r.inputs(), modulus=list(r.take(r.length())), output=r.u32())
    elif name == "BigIntToLeBytes":
        f.update(input=r.u32(), outputs=r.witnesses())
    elif name == "Poseidon2Permutation":
        f.update(inputs=r.inputs(), outputs=r.witnesses(), len=r.u32())
    elif name == "Sha256Compression":
        f.update(inputs=r.inputs(16), hash_values=r.inputs(8), outputs=r.witnesses(8))
    return f


def _brillig_input(r):
    kind = r.tag(["Single", "Array", "MemoryArray"], "BrilligInputs")
    if kind == "Single":
        return (kind, r.expression())
    if kind == "Array":
        return (kind, r.vec(r.expression))
    return (kind, r.u32())


def _brillig_output(r):
    kind = r.tag(["Simple", "Array"], "BrilligOutputs")
    return (kind, r.witness() if kind == "Simple" else r.witnesses())


def _opcode(r):
    kind = r.tag(OPCODES, "Opcode")
    if kind == "AssertZero":
        return (kind, r.expression())
    if kind == "BlackBoxFuncCall":
        return (kind, _black_box(r))
    if kind == "Directive":
        r.tag(["ToLeRadix"], "Directive")
        return (kind, {"name": "ToLeRadix", "a": r.expression(), "b": r.witnesses(), "radix": r.u32()})
    if kind == "MemoryOp":
        block = r.u32()
        op = {"operation": r.expression(), "index": r.expression(), "value": r.expression()}
        return (kind, {"block_id": block, "op": op, "predicate": r.option(r.expression)})
    if kind == "MemoryInit":
        return (kind, {"block_id": r.u32(), "init": r.witnesses(), "block_type": r.tag(["Memory", "CallData", "ReturnData"], "BlockType")})
    if kind == "BrilligCall":
        return (kind, {"id": r.u32(), "inputs": r.vec(lambda: _brillig_input(r)), "outputs": r.vec(lambda: _brillig_output(r)),
                       "predicate": r.option(r.expression)})
    return (kind, {"id": r.u32(), "inputs": r.witnesses(), "outputs": r.witnesses(), "predicate": r.option(r.expression)})


def _assert_message(r):
    loc = r.tag(["Acir", "Brillig"], "OpcodeLocation")
    location = (loc, r.u64()) if loc == "Acir" else (loc, r.u64(), r.u64())
    kind = r.tag(["StaticString", "Dynamic"], "AssertionPayload")
    if kind == "StaticString":
        return (location, (kind, r.string()))

    def expr_or_mem():
        k = r.tag(["Expression", "Memory"], "ExpressionOrMemory")
        return (k, r.expression() if k == "Expression" else r.u32())

    return (location, (kind, r.u64(), r.vec(expr_or_mem)))


def _circuit(r):
    c = {"current_witness_index": r.u32(), "opcodes": r.vec(lambda: _opcode(r))}
    width = r.tag(["Unbounded", "Bounded"], "ExpressionWidth")
    c["expression_width"] = (width, r.u64()) if width == "Bounded" else (width,)
    c["private_parameters"] = r.witnesses()
    c["public_parameters"] = r.witnesses()
    c["return_values"] = r.witnesses()
    c["assert_messages"] = r.vec(lambda: _assert_message(r))
    c["recursive"] = r.boolean()
    return c


def _gunzip(data, what):
    try:
        return gzip.decompress(bytes(data))
    except (OSError, EOFError, ValueError) as e:
        raise AcirFormatError("%s is not gzip data: %s" % (what, e))


def deserialize_program(bytecode):
    """``Program::deserialize_program``: {"functions": [circuit, ...]}.  The Brillig bytecode of the
    unconstrained functions that follows the circuits is not parsed: the backend ignores BrilligCall
    (circuit_translation/mod.rs:98-103) and nothing on the proving path reads it."""
    r = _In(_gunzip(bytecode, "program bytecode"))
    return {"functions": r.vec(lambda: _circuit(r)), "unconstrained_functions_offset": r.at}


def deserialize_program_within_file_path(path):
    """noir_and_plonky2_serialization.rs:42-58."""
    with open(path) as f:
        try:
            artefact = json.load(f)
        except ValueError as e:
            raise AcirFormatError("There was a problem parsing the json program: %s" % e)
    code = artefact.get("bytecode") if isinstance(artefact, dict) else None
    if not isinstance(code, str):
        raise AcirFormatError("Expected a different circuit format")
    try:
        raw = base64.b64decode(code, validate=True)
    except ValueError as e:
        raise AcirFormatError("There was a problem decoding the program from base 64: %s" % e)
    return deserialize_program(raw)


def deserialize_witnesses(data):
    """``WitnessStack::try_from``: [{"index": function index, "witness": {witness: value mod p}}, ...], bottom of
    the stack first (the backend pops the last one, actions/prove_action.rs:108)."""
    r = _In(_gunzip(data, "witness file"))
    stack = r.vec(lambda: {"index": r.u32(), "witness": dict(r.vec(lambda: (r.witness(), r.field())))})
    if r.at != len(r.d):
        raise AcirFormatError("trailing bytes after the witness stack")
    return stack


def deserialize_witnesses_within_file_path(path):
    """noir_and_plonky2_serialization.rs:60-64."""
    with open(path, "rb") as f:
        return deserialize_witnesses(f.read())


def to_translator_opcodes(circuit):
    """The opcode list ``translate.CircuitBuilderFromAcirToPlonky2.translate_circuit`` takes, for the opcodes it
    restates (AssertZero, RANGE, AND, XOR, Sha256Compression); BrilligCall and Directive are dropped as the
    reference drops them (mod.rs:98-104); anything else raises NotImplementedError naming the opcode."""
    out = []
    for kind, body in circuit["opcodes"]:
        if kind == "AssertZero":
            out.append(("assert_zero", body["mul_terms"], body["linear_combinations"], body["q_c"]))
        elif kind in ("BrilligCall", "Directive"):
            continue
        elif kind == "BlackBoxFuncCall":
            n = body["name"]
            if n == "RANGE":
                out.append(("range", body["input"][0], body["input"][1]))
            elif n in ("AND", "XOR"):
                out.append((n.lower(), body["lhs"][0], body["rhs"][0], body["output"], body["lhs"][1]))
            elif n == "Sha256Compression":
                out.append(("sha256_compression", [w for w, _ in body["inputs"]], [w for w, _ in body["hash_values"]], body["outputs"]))
            else:
                raise NotImplementedError("BlackBoxFuncCall::%s" % n)
        else:
            raise NotImplementedError("Opcode::%s" % kind)
    return out


# ---- the inverse functions (nargo's side): used by the tests to make program / witness files -----------------
class _Out:
    def __init__(self):
        self.b = bytearray()

    def u8(self, x):
        self.b.append(x)

    def u32(self, x):
        self.b += struct.pack("<I", x)

    def u64(self, x):
        self.b += struct.pack("<Q", x)

    def string(self, s):
        e = s.encode()
        self.u64(len(e))
        self.b += e

    def field(self, v, width=32):
        self.string("%0*x" % (2 * width, v % P))

    def vec(self, items, put):
        self.u64(len(items))
        for it in items:
            put(it)

    def option(self, v, put):
        self.u8(0 if v is None else 1)
        if v is not None:
            put(v)

    def expression(self, e):
        self.vec(e["mul_terms"], lambda t: (self.field(t[0]), self.u32(t[1]), self.u32(t[2])))
        self.vec(e["linear_combinations"], lambda t: (self.field(t[0]), self.u32(t[1])))
        self.field(e["q_c"])

    def function_input(self, fi):
        self.u32(fi[0])
        self.u32(fi[1])


def _put_black_box(o, f):
    n = f["name"]
    o.u32(BLACK_BOX.index(n))
    if n in ("AND", "XOR"):
        o.function_input(f["lhs"]); o.function_input(f["rhs"]); o.u32(f["output"])
    elif n == "RANGE":
        o.function_input(f["input"])
    elif n == "Sha256Compression":
        assert len(f["inputs"]) == 16 and len(f["hash_values"]) == 8 and len(f["outputs"]) == 8
        for fi in f["inputs"] + f["hash_values"]:
            o.function_input(fi)
        for w in f["outputs"]:
            o.u32(w)
    elif n in ("EcdsaSecp256k1", "EcdsaSecp256r1"):
        for key, cnt in (("public_key_x", 32), ("public_key_y", 32), ("signature", 64), ("hashed_message", 32)):
            assert len(f[key]) == cnt
            for fi in f[key]:
                o.function_input(fi)
        o.u32(f["output"])
    elif n in ("SHA256", "Blake2s", "Blake3"):
        o.vec(f["inputs"], o.function_input)
        for w in f["outputs"]:
            o.u32(w)
    else:
        raise NotImplementedError("writer: BlackBoxFuncCall::%s" % n)


def _put_opcode(o, op):
    kind, body = op
    o.u32(OPCODES.index(kind))
    if kind == "AssertZero":
        o.expression(body)
    elif kind == "BlackBoxFuncCall":
        _put_black_box(o, body)
    elif kind == "Directive":
        o.u32(0); o.expression(body["a"]); o.vec(body["b"], o.u32); o.u32(body["radix"])
    elif kind == "MemoryOp":
        o.u32(body["block_id"])
        for k in ("operation", "index", "value"):
            o.expression(body["op"][k])
        o.option(body["predicate"], o.expression)
    elif kind == "MemoryInit":
        o.u32(body["block_id"]); o.vec(body["init"], o.u32); o.u32(["Memory", "CallData", "ReturnData"].index(body["block_type"]))
    elif kind == "BrilligCall":
        o.u32(body["id"])

        def put_in(x):
            o.u32(["Single", "Array", "MemoryArray"].index(x[0]))
            if x[0] == "Single":
                o.expression(x[1])
            elif x[0] == "Array":
                o.vec(x[1], o.expression)
            else:
                o.u32(x[1])

        def put_out(x):
            o.u32(["Simple", "Array"].index(x[0]))
            if x[0] == "Simple":
                o.u32(x[1])
            else:
                o.vec(x[1], o.u32)

        o.vec(body["inputs"], put_in); o.vec(body["outputs"], put_out); o.option(body["predicate"], o.expression)
    else:
        o.u32(body["id"]); o.vec(body["inputs"], o.u32); o.vec(body["outputs"], o.u32); o.option(body["predicate"], o.expression)


def serialize_program(functions, unconstrained_functions=0):
    """``Program::serialize_program`` for programs without Brillig bytecode (an empty
    ``unconstrained_functions`` vector)."""
    assert unconstrained_functions == 0
    o = _Out()
    o.u64(len(functions))
    for c in functions:
        o.u32(c.get("current_witness_index", 0))
        o.vec(c["opcodes"], lambda op: _put_opcode(o, op))
        width = c.get("expression_width", ("Unbounded",))
        o.u32(["Unbounded", "Bounded"].index(width[0]))
        if width[0] == "Bounded":
            o.u64(width[1])
        for key in ("private_parameters", "public_parameters", "return_values"):
            o.vec(sorted(c.get(key, [])), o.u32)
        msgs = c.get("assert_messages", [])
        o.u64(len(msgs))
        for location, payload in msgs:
            o.u32(["Acir", "Brillig"].index(location[0]))
            for x in location[1:]:
                o.u64(x)
            assert payload[0] == "StaticString"
            o.u32(0); o.string(payload[1])
        o.u8(1 if c.get("recursive", False) else 0)
    o.u64(0)
    return gzip.compress(bytes(o.b), mtime=0)


def program_json(functions):
    """The ``target/<name>.json`` artefact, reduced to what the backend reads."""
    return json.dumps({"noir_version": "0.31.0", "bytecode": base64.b64encode(serialize_program(functions)).decode()})


def serialize_witness_stack(stack):
    o = _Out()
    o.u64(len(stack))
    for item in stack:
        o.u32(item["index"])
        pairs = sorted(item["witness"].items())
        o.vec(pairs, lambda kv: (o.u32(kv[0]), o.field(kv[1])))
    return gzip.compress(bytes(o.b), mtime=0)
