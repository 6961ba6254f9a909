"""`mux_circuits::MuxCircuit` on the GPU gate-graph executor (BASELINE configs 3 and 5).

The reference describes integer arithmetic as multiplexer trees over the input bits
(mux_circuits/src/lib.rs:58-170: `MuxOp::{One, Zero, Mux, Variable(i), Output(i)}` joined by
`MuxEdgeInfo::{Low, High, Select, Output}` edges) and lowers them into its `FheCircuit` with
`FheCircuit::insert_mux_circuit` (parasol_runtime/src/fhe_circuit.rs:274-420): every `Mux` becomes an
`FheOp::CMux` whose selector is the GGSW of an input bit, `One` / `Zero` become trivial GLWEs.  Its
multiplier blocks are BDD-derived circuits shipped as bincode blobs
(`mux_circuits::mul::unsigned_multiplier`, mux_circuits/src/mul.rs:62-69: `bincode::deserialize` of
`data/multiplier-n8-m8`, `-n16-m16`), which this module reads:

    bincode 1.x default options (fixed-width little-endian integers, u64 lengths, u32 enum variant indices)
    of  struct MuxCircuit { graph: StableGraph<MuxOp, MuxEdgeInfo>, inputs: Vec<NodeIndex> }
    with petgraph 0.7's serde form of a StableGraph:
        nodes: Vec<MuxOp>                 (u64 n, then n x { u32 variant [, u32 payload for Variable / Output] })
        node_holes: Vec<NodeIndex<u32>>   (u64 n, n x u32)
        edge_property: enum               (u32; 1 = Directed)
        edges: Vec<Option<(u32 source, u32 target, MuxEdgeInfo)>>   (u64 n, n x { u8 tag [, u32, u32, u32] })
    inputs: u64 n, n x u32                (node index of Variable(i), in input order)

No FHE happens here: this is graph plumbing over `spf_amd.FheCircuit` (and a plaintext evaluator that the
tests use to pin the parser against integer multiplication).
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

ONE, ZERO, MUX, VARIABLE, OUTPUT = 0, 1, 2, 3, 4      # MuxOp variant indices (declaration order, lib.rs:58-95)
LOW, HIGH, SELECT, OUT_EDGE = 0, 1, 2, 3              # MuxEdgeInfo variant indices (lib.rs:99-111)


class MuxFormatError(ValueError):
    pass


@dataclass
class MuxCircuit:
    ops: List[Tuple[int, Optional[int]]]               # (variant, payload) per node
    low: List[int]                                     # per node: source of its Low / High / Select / Output edge, or -1
    high: List[int]
    select: List[int]
    out_src: List[int]
    inputs: List[int]                                  # node index of input i

    @property
    def n_inputs(self) -> int:
        return len(self.inputs)

    @property
    def outputs(self) -> List[int]:
        """node index of Output(i), by i"""
        outs = sorted((p, i) for i, (v, p) in enumerate(self.ops) if v == OUTPUT)
        return [i for _, i in outs]

    def metrics(self):
        """`MuxCircuit::metrics` (lib.rs:199-220)"""
        return {"mux_gates": sum(v == MUX for v, _ in self.ops), "inputs": sum(v == VARIABLE for v, _ in self.ops),
                "outputs": sum(v == OUTPUT for v, _ in self.ops)}

    def topological_muxes(self) -> List[int]:
        """Mux nodes, every one after the Mux nodes its Low / High edges come from (iterative DFS)."""
        order, state = [], {}
        for root in self.outputs:
            stack = [(self.out_src[root], False)]
            while stack:
                n, done = stack.pop()
                if self.ops[n][0] != MUX or state.get(n) == 2:
                    continue
                if done:
                    state[n] = 2
                    order.append(n)
                    continue
                if state.get(n) == 1:
                    raise MuxFormatError("cycle in the multiplexer graph")
                state[n] = 1
                stack.append((n, True))
                stack.append((self.low[n], False))
                stack.append((self.high[n], False))
        return order

    def depth(self) -> int:
        d = {}
        for n in self.topological_muxes():
            d[n] = 1 + max(d.get(self.low[n], 0), d.get(self.high[n], 0))
        return max((d.get(self.out_src[o], 0) for o in self.outputs), default=0)


DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def unsigned_multiplier(n: int, m: int) -> "MuxCircuit":
    """`mux_circuits::mul::unsigned_multiplier(n, m)` (mux_circuits/src/mul.rs:62-69): the reference ships these
    blocks as precomputed bincode blobs and `include_bytes!`es them; spf_amd/data/ holds the same files (README there)."""
    path = os.path.join(DATA_DIR, f"mux_multiplier_n{n}_m{m}.bincode")
    if not os.path.exists(path):
        raise FileNotFoundError(f"no precomputed multiplier block for {n} x {m} bits ({path})")
    with open(path, "rb") as f:
        return parse_mux_circuit(f.read())


def parse_mux_circuit(blob: bytes) -> MuxCircuit:
    off = 0

    def take(fmt, size):
        nonlocal off
        if off + size > len(blob):
            raise MuxFormatError("truncated MuxCircuit")
        (v,) = struct.unpack_from(fmt, blob, off)
        off += size
        return v

    u64, u32, u8 = (lambda: take("<Q", 8)), (lambda: take("<I", 4)), (lambda: take("<B", 1))
    n = u64()
    if n > len(blob):
        raise MuxFormatError("node count exceeds the buffer")
    ops = []
    for _ in range(n):
        v = u32()
        if v > OUTPUT:
            raise MuxFormatError(f"unknown MuxOp variant {v}")
        ops.append((v, u32() if v in (VARIABLE, OUTPUT) else None))
    holes = [u32() for _ in range(u64())]
    if holes:
        raise MuxFormatError("StableGraph with vacant node slots: not produced by the reference's circuits")
    if u32() != 1:
        raise MuxFormatError("the graph must be directed")
    low, high, select, out_src = ([-1] * n for _ in range(4))
    slot = {LOW: low, HIGH: high, SELECT: select, OUT_EDGE: out_src}
    for _ in range(u64()):
        if u8():
            s, t, kind = u32(), u32(), u32()
            if s >= n or t >= n or kind not in slot:
                raise MuxFormatError("edge out of range")
            if slot[kind][t] != -1:
                raise MuxFormatError("a node has two edges of the same kind")
            slot[kind][t] = s
    inputs = [u32() for _ in range(u64())]
    for i, node in enumerate(inputs):
        if node >= n or ops[node] != (VARIABLE, i):
            raise MuxFormatError("inputs[i] must be the node Variable(i)")
    for i, (v, _) in enumerate(ops):
        if v == MUX and (low[i] < 0 or high[i] < 0 or select[i] < 0):
            raise MuxFormatError("a Mux needs one Low, one High and one Select edge")
        if v == MUX and ops[select[i]][0] != VARIABLE:
            raise MuxFormatError("Select lines must come from input variables")
        if v == OUTPUT and out_src[i] < 0:
            raise MuxFormatError("an Output needs its edge")
    return MuxCircuit(ops, low, high, select, out_src, inputs)


def evaluate_plain(c: MuxCircuit, bits: Sequence[int]) -> List[int]:
    """the circuit on plaintext bits (mux = low when select is 0, high when it is 1; lib.rs:71-79)"""
    if len(bits) != c.n_inputs:
        raise ValueError("wrong number of input bits")
    val = {}
    for i, (v, p) in enumerate(c.ops):
        if v == ONE:
            val[i] = 1
        elif v == ZERO:
            val[i] = 0
        elif v == VARIABLE:
            val[i] = int(bits[p]) & 1
    for n in c.topological_muxes():
        val[n] = val[c.high[n]] if val[c.select[n]] else val[c.low[n]]
    return [val[c.out_src[o]] for o in c.outputs]


def insert_mux_circuit(graph, c: MuxCircuit, ggsw_inputs: Sequence[int], add_cmux: Optional[Callable] = None) -> List[int]:
    """`FheCircuit::insert_mux_circuit(.., MuxMode::Glwe)` (fhe_circuit.rs:274-420) on a `spf_amd.FheCircuit`:
    `ggsw_inputs[i]` is the graph node holding the GGSW of input bit i; returns the GLWE nodes of the outputs,
    by output index."""
    from .graph import FheOp, ValueKind
    if len(ggsw_inputs) != c.n_inputs:
        raise ValueError("one GGSW node per circuit input")
    node = {}
    for i, (v, p) in enumerate(c.ops):
        if v == ONE:
            node[i] = graph.add_trivial(ValueKind.GLWE1, 1)
        elif v == ZERO:
            node[i] = graph.add_trivial(ValueKind.GLWE1, 0)
    for n in c.topological_muxes():
        sel = ggsw_inputs[c.ops[c.select[n]][1]]
        node[n] = graph.add_op(FheOp.CMux, [sel, node[c.low[n]], node[c.high[n]]])
    return [node[c.out_src[o]] for o in c.outputs]


# ---- circuits the reference GENERATES (no blob): reduced ordered BDDs of the outputs, one Mux per BDD node ----------
#
# `mux_circuits::add::ripple_carry_adder` (mux_circuits/src/add.rs:13-58) builds the BDD of every sum bit over the
# variables [carry-in,] a0, b0, a1, b1, ... with a BDD library and converts them with `MuxCircuit::from(&[Bdd])`
# (lib.rs:358-445): per output, ONE Mux per internal BDD node (select = the node's variable, low / high = its
# children), terminals = the shared One / Zero, nothing shared between outputs.  A reduced ordered BDD is canonical
# for a function and a variable order, so any correct ROBDD construction yields the same multiplexer DAG (same gate
# count, same depth) as the reference's library; the little engine below is that construction.

class _Robdd:
    """reduced ordered BDDs with a unique table; node ids 0 / 1 are the terminals"""

    def __init__(self, n_vars: int):
        self.n_vars = n_vars
        self.nodes = [(n_vars, 0, 0), (n_vars, 1, 1)]
        self.unique = {}
        self.memo = {}

    def mk(self, var: int, lo: int, hi: int) -> int:
        if lo == hi:
            return lo
        key = (var, lo, hi)
        n = self.unique.get(key)
        if n is None:
            n = len(self.nodes)
            self.nodes.append(key)
            self.unique[key] = n
        return n

    def var(self, i: int) -> int:
        return self.mk(i, 0, 1)

    def apply(self, op: str, f: int, g: int) -> int:
        if f < 2 and g < 2:
            return {"and": f & g, "or": f | g, "xor": f ^ g}[op]
        key = (op, f, g) if f <= g else (op, g, f)
        r = self.memo.get(key)
        if r is not None:
            return r
        vf, vg = self.nodes[f][0], self.nodes[g][0]
        v = min(vf, vg)
        f0, f1 = (self.nodes[f][1], self.nodes[f][2]) if vf == v else (f, f)
        g0, g1 = (self.nodes[g][1], self.nodes[g][2]) if vg == v else (g, g)
        r = self.mk(v, self.apply(op, f0, g0), self.apply(op, f1, g1))
        self.memo[key] = r
        return r


def mux_circuit_from_bdds(bdd: _Robdd, roots: Sequence[int]) -> MuxCircuit:
    """`MuxCircuit::from(&[Bdd])` (lib.rs:358-445)"""
    ops: List[Tuple[int, Optional[int]]] = []
    low: List[int] = []
    high: List[int] = []
    select_var: List[int] = []
    out_src: List[int] = []

    def add(op, payload=None):
        ops.append((op, payload))
        low.append(-1), high.append(-1), select_var.append(-1), out_src.append(-1)
        return len(ops) - 1

    terminal = {}
    for i, root in enumerate(roots):
        local = {}
        order, stack, seen = [], [(root, False)], set()
        while stack:                                         # children before parents
            n, done = stack.pop()
            if n < 2 or (n in seen and not done):
                continue
            if done:
                order.append(n)
                continue
            seen.add(n)
            stack.append((n, True))
            stack.append((bdd.nodes[n][1], False))
            stack.append((bdd.nodes[n][2], False))

        def node_of(n):
            if n < 2:
                if n not in terminal:
                    terminal[n] = add(ONE if n else ZERO)
                return terminal[n]
            return local[n]

        for n in order:
            v, lo, hi = bdd.nodes[n]
            m = add(MUX)
            low[m], high[m], select_var[m] = node_of(lo), node_of(hi), v
            local[n] = m
        o = add(OUTPUT, i)
        out_src[o] = node_of(root)
    inputs = [add(VARIABLE, v) for v in range(bdd.n_vars)]
    select = [inputs[v] if v >= 0 else -1 for v in select_var] + [-1] * (len(ops) - len(select_var))
    return MuxCircuit(ops, low, high, select[:len(ops)], out_src, inputs)


def ripple_carry_adder(n: int, m: int, cin: bool) -> MuxCircuit:
    """`mux_circuits::add::ripple_carry_adder` (add.rs:13-58): an n-bit plus an m-bit integer (plus a carry-in),
    max(n, m) + 1 output bits (the top one is the carry out).  Inputs: [carry-in,] a0, b0, a1, b1, ... until the
    shorter operand is exhausted, then the rest of the longer one."""
    if n <= 0 or m <= 0:
        raise ValueError("operand widths must be positive")
    lo_len, hi_len, off = min(n, m), max(n, m), int(bool(cin))
    bdd = _Robdd(n + m + off)
    carry = bdd.var(0) if cin else 0
    sums = []
    for i in range(lo_len):
        a, b = bdd.var(off + 2 * i), bdd.var(off + 2 * i + 1)
        axb = bdd.apply("xor", a, b)
        sums.append(bdd.apply("xor", carry, axb))
        carry = bdd.apply("or", bdd.apply("and", axb, carry), bdd.apply("and", a, b))
    for i in range(hi_len - lo_len):
        a = bdd.var(2 * lo_len + i + off)
        sums.append(bdd.apply("xor", carry, a))
        carry = bdd.apply("and", a, carry)
    sums.append(carry)
    return mux_circuit_from_bdds(bdd, sums)


def common_subexpression_elimination(c: MuxCircuit) -> MuxCircuit:
    """`MuxCircuit::optimize` (lib.rs:274-279 -> opt.rs:328-415): multiplexers with the same select, low and high
    operands are one multiplexer; applied in topological order so that merged children merge their parents."""
    canon = {}
    for i, (v, _) in enumerate(c.ops):
        if v != MUX:
            canon[i] = i
    seen = {}
    for n in c.topological_muxes():
        key = (c.select[n], canon[c.low[n]], canon[c.high[n]])
        canon[n] = seen.setdefault(key, n)
    keep = [i for i, (v, _) in enumerate(c.ops) if v != MUX or canon[i] == i]
    new_index = {old: k for k, old in enumerate(keep)}
    remap = lambda x: -1 if x < 0 else new_index[canon[x]]
    return MuxCircuit([c.ops[i] for i in keep], [remap(c.low[i]) for i in keep], [remap(c.high[i]) for i in keep],
                      [remap(c.select[i]) for i in keep], [remap(c.out_src[i]) for i in keep], [new_index[i] for i in c.inputs])


CIRCUIT_CUTOFF = 16     # mul.rs:243


def partition_integer(n: int) -> Tuple[int, int]:
    """mul.rs:245-260: (low, high) word lengths; no split up to the cutoff"""
    return (n, 0) if n <= CIRCUIT_CUTOFF else ((n + 1) // 2, n // 2)


def encode_gradeschool_reduction(n: int, m: int, lo_lo, lo_hi, hi_lo, hi_hi) -> list:
    """mul.rs:262-388: the order in which the bits of the four partial products a_lo*b_lo, a_lo*b_hi, a_hi*b_lo,
    a_hi*b_hi enter the reduction circuit (by the column they are added in)."""
    a_lo, a_hi = partition_integer(n)
    b_lo, b_hi = partition_integer(m)
    assert len(lo_lo) == a_lo + b_lo and len(lo_hi) == a_lo + b_hi and len(hi_lo) == a_hi + b_lo and len(hi_hi) == a_hi + b_hi
    assert a_lo >= b_lo and a_hi <= a_lo and b_hi <= b_lo
    out, o = [], {"ll": 0, "hl": 0, "lh": 0, "hh": 0}
    src = {"ll": lo_lo, "hl": hi_lo, "lh": lo_hi, "hh": hi_hi}
    for run, names in ((b_lo, ("ll",)), (a_lo - b_lo, ("ll", "lh")), (b_lo, ("ll", "hl", "lh")), (b_hi, ("hl", "lh", "hh")),
                       (a_hi - b_hi, ("hl", "hh")), (b_hi, ("hh",))):
        for i in range(run):
            for k in names:
                out.append(src[k][o[k] + i])
        for k in names:
            o[k] += run
    return out


def gradeschool_reduce(n: int, m: int) -> MuxCircuit:
    """`gradeschool_reduce_impl` (mul.rs:390-590): the 4-operand shifted addition of the partial products as BDDs over
    the encoded bit order, one multiplexer per BDD node, then `optimize`.  (The reference ships only the 64 x 64
    instance as a blob and generates the others like this at run time.)"""
    assert n >= m
    a_lo, a_hi = partition_integer(n)
    b_lo, b_hi = partition_integer(m)
    bdd = _Robdd(2 * (n + m))
    V = [bdd.var(i) for i in range(2 * (n + m))]
    xor, land, lor = (lambda f, g: bdd.apply("xor", f, g)), (lambda f, g: bdd.apply("and", f, g)), (lambda f, g: bdd.apply("or", f, g))
    neg = lambda f: bdd.apply("xor", f, 1)

    def exactly(bits, k):                      # n_bits_are_true (mul.rs:213-240)
        import itertools
        res = 0
        for chosen in itertools.combinations(range(len(bits)), k):
            clause = 1
            for i, x in enumerate(bits):
                clause = land(clause, x if i in chosen else neg(x))
            res = lor(res, clause)
        return res

    result = [0] * (n + m)
    c0 = c1 = c2 = 0
    for i in range(n + m):                     # section 1 as the reference writes it: every output starts as input i
        result[i] = V[i]
    i_off, o_off = b_lo, b_lo
    for i in range(a_lo - b_lo):               # section 2: two operands + one carry
        a, b = V[i_off + 2 * i], V[i_off + 2 * i + 1]
        ops = [a, b, c0]
        result[o_off + i] = xor(xor(a, b), c0)
        c0 = lor(exactly(ops, 2), exactly(ops, 3))
    i_off += 2 * (a_lo - b_lo)
    o_off += a_lo - b_lo
    for i in range(b_lo + b_hi):               # sections 3, 4: three operands + two carries
        a, b, c = V[i_off + 3 * i], V[i_off + 3 * i + 1], V[i_off + 3 * i + 2]
        result[o_off + i] = xor(xor(xor(xor(a, b), c), c0), c1)
        ops = [a, b, c, c0, c1]
        two, three, four, five = (exactly(ops, k) for k in (2, 3, 4, 5))
        c0 = lor(two, three)
        c1 = c2
        c2 = lor(four, five)
    i_off += 3 * (b_lo + b_hi)
    o_off += b_lo + b_hi
    for i in range(a_hi - b_hi):               # section 5: two operands + two carries
        a, b = V[i_off + 2 * i], V[i_off + 2 * i + 1]
        ops = [a, b, c0, c1]
        two, three, four = (exactly(ops, k) for k in (2, 3, 4))
        result[o_off + i] = xor(xor(xor(a, b), c0), c1)
        c0 = lor(two, three)
        c1 = c2
        c2 = four
    i_off += 2 * (a_hi - b_hi)
    o_off += a_hi - b_hi
    for i in range(b_hi):                      # section 6: carries ripple into a_hi * b_hi
        a = V[i_off + i]
        if i < 2:
            result[o_off + i] = xor(xor(a, c0), c1)
            ops = [a, c0, c1]
            c0 = lor(exactly(ops, 2), exactly(ops, 3))
            if i == 0:
                c1 = c2
        else:
            result[o_off + i] = xor(a, c0)
            c0 = land(a, c0)
    return common_subexpression_elimination(mux_circuit_from_bdds(bdd, result))


class PlainBuilder:
    """plaintext stand-in for the gate graph (tests): nodes are bits"""

    def insert(self, circuit: MuxCircuit, inputs):
        return evaluate_plain(circuit, inputs)

    def to_ggsw(self, bit):
        return bit


class GraphBuilder:
    """the same interface over `spf_amd.FheCircuit`: `insert` = insert_mux_circuit (MuxMode::Glwe), `to_ggsw` =
    insert_ciphertext_conversion(L1Glwe -> L1Ggsw) = SampleExtract(0), KeyswitchL1toL0, CircuitBootstrap
    (fhe_circuit.rs:563-622)"""

    def __init__(self, graph):
        self.graph = graph

    def insert(self, circuit: MuxCircuit, ggsw_nodes):
        return insert_mux_circuit(self.graph, circuit, ggsw_nodes)

    def to_ggsw(self, glwe_node):
        from .graph import FheOp
        x = self.graph.add_op(FheOp.SampleExtract, [glwe_node], 0)
        x = self.graph.add_op(FheOp.KeyswitchL1toL0, [x])
        return self.graph.add_op(FheOp.CircuitBootstrap, [x])


def append_uint_multiply(builder, a: Sequence, b: Sequence, blocks: Callable[[int, int], MuxCircuit]) -> list:
    """`mul_impl` (parasol_runtime/src/circuits/mul.rs:90-200): recursive gradeschool multiplication of two unsigned
    integers given as GGSW bit nodes (LSB first); returns the len(a) + len(b) product bits as GLWE nodes.
    `blocks(n, m)` supplies `unsigned_multiplier(n, m)` (the reference loads 8 x 8, 16 x 16 from blobs)."""
    if len(a) < len(b):
        a, b = b, a
    a_lo_len, a_hi_len = partition_integer(len(a))
    b_lo_len, b_hi_len = partition_integer(len(b))
    a_lo, a_hi, b_lo, b_hi = a[:a_lo_len], a[a_lo_len:], b[:b_lo_len], b[b_lo_len:]
    if a_hi_len == 0 and b_hi_len == 0:
        return builder.insert(blocks(len(a), len(b)), list(a) + list(b))
    if b_hi_len == 0:
        ll = append_uint_multiply(builder, a_lo, b_lo, blocks)
        hl = append_uint_multiply(builder, a_hi, b_lo, blocks)
        adder = ripple_carry_adder(b_lo_len, a_hi_len + b_lo_len, False)
        lo, hi = ll[:a_lo_len], ll[a_lo_len:]
        ins = [x for pair in zip(hi, hl[:a_lo_len]) for x in pair] + list(hl[a_lo_len:])
        return lo + builder.insert(adder, [builder.to_ggsw(x) for x in ins])
    ll = append_uint_multiply(builder, a_lo, b_lo, blocks)
    lh = append_uint_multiply(builder, a_lo, b_hi, blocks)
    hl = append_uint_multiply(builder, a_hi, b_lo, blocks)
    hh = append_uint_multiply(builder, a_hi, b_hi, blocks)
    bits = encode_gradeschool_reduction(len(a), len(b), ll, lh, hl, hh)
    return builder.insert(gradeschool_reduce(len(a), len(b)), [builder.to_ggsw(x) for x in bits])
