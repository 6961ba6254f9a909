"""Host-side mirror of ``FheCircuit`` + ``CircuitProcessor::run_graph_blocking`` over the graph
executor of the C ABI (``spf_graph_*``, include/spf_hip.h; spf_amd/csrc/spf_graph.hpp).

The reference builds a petgraph DAG of ``FheOp`` nodes joined by typed ``FheEdge``s
(parasol_runtime/src/fhe_circuit.rs:34-205) and runs it with one rayon task per node
(circuit_processor/mod.rs:130-253, 573-623).  Here the same DAG is handed to the library, which runs
it level by level as batched launches with every intermediate in HBM.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import List, Sequence

import numpy as np

from ._ffi import Engine, SpfError, _ptr


class ValueKind(enum.IntEnum):
    LWE0 = 0
    LWE1 = 1
    GLWE1 = 2
    GGSW1 = 3
    GLEV1 = 4


class FheOp(enum.IntEnum):
    """The computing variants of ``FheOp`` (fhe_circuit.rs:65-126); inputs, outputs and constants have
    their own methods on :class:`FheCircuit`."""
    SampleExtract = 0
    KeyswitchL1toL0 = 1
    Not = 2
    GlweAdd = 3
    CMux = 4
    GlevCMux = 5
    MultiplyGgswGlwe = 6
    CircuitBootstrap = 7
    SchemeSwitch = 8
    MulXN = 9


class FheCircuit:
    def __init__(self, engine: Engine):
        self._eng = engine
        self._lib = engine._lib
        h = C.c_void_p()
        from ._ffi import Group
        if isinstance(engine, Group):   # a job of the group: placed on a member when it is run (Group.run_graphs / run())
            engine._ck(engine._raw.spf_group_graph_create(engine._h, C.byref(h)))
            self._lib = engine._raw
        else:
            engine._ck(self._lib.spf_graph_create(engine._h, C.byref(h)))
        self._g = h
        self._keep: List[np.ndarray] = []   # input / output buffers the library reads and writes at run()

    def close(self):
        if getattr(self, "_g", None):
            self._lib.spf_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _words(self, kind: ValueKind) -> int:
        p = self._eng.params
        return {ValueKind.LWE0: p.lwe0_words, ValueKind.LWE1: p.lwe1_words, ValueKind.GLWE1: p.glwe_words,
                ValueKind.GGSW1: 2 * p.cbs_ggsw_complex,
                ValueKind.GLEV1: p.cbs_radix_count * p.glwe_words}[ValueKind(kind)]

    # FheOp::Input* — the array is read at every run(); change its contents to re-run on new data
    def add_input(self, kind: ValueKind, value: np.ndarray) -> int:
        a = np.ascontiguousarray(value)
        if a.nbytes != self._words(kind) * 8:
            raise SpfError(1, f"input of kind {ValueKind(kind).name} must have {self._words(kind) * 8} bytes, got {a.nbytes}")
        self._keep.append(a)
        node = C.c_uint32()
        self._eng._ck(self._lib.spf_graph_add_input(self._g, int(kind), _ptr(a), C.byref(node)))
        return node.value

    # FheOp::{Zero,One}{Lwe0,Glwe1,Glev1,Ggsw1} (fhe_circuit.rs:96-116)
    def add_trivial(self, kind: ValueKind, bit: int) -> int:
        node = C.c_uint32()
        self._eng._ck(self._lib.spf_graph_add_trivial(self._g, int(kind), bit, C.byref(node)))
        return node.value

    def add_op(self, op: FheOp, inputs: Sequence[int], param: int = 0) -> int:
        arr = (C.c_uint32 * len(inputs))(*inputs)
        node = C.c_uint32()
        self._eng._ck(self._lib.spf_graph_add_op(self._g, int(op), arr, len(inputs), param, C.byref(node)))
        return node.value

    # FheOp::Output* — returns the array run() fills
    def add_output(self, node: int, kind: ValueKind) -> np.ndarray:
        dtype = np.complex128 if ValueKind(kind) == ValueKind.GGSW1 else np.uint64
        out = np.zeros(self._words(kind) * 8 // np.dtype(dtype).itemsize, dtype=dtype)
        self._keep.append(out)
        self._eng._ck(self._lib.spf_graph_add_output(self._g, node, _ptr(out)))
        return out

    # CircuitProcessor::run_graph_blocking
    def run(self):
        self._eng._ck(self._lib.spf_graph_run(self._g))

    def member(self) -> int:
        """the member of the group the graph last ran on (0 for a graph of one engine)"""
        return int(self._lib.spf_graph_member(self._g))

    def stats(self):
        n, lv, la = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._eng._ck(self._lib.spf_graph_stats(self._g, C.byref(n), C.byref(lv), C.byref(la)))
        return {"nodes": n.value, "levels": lv.value, "launches": la.value}


class RecordedCircuit:
    """The builder calls of :class:`FheCircuit` kept as plain arrays, not bound to an executor: the same DAG can then be
    lowered into a gate graph (`lower`) or walked node by node the way the reference's `CircuitProcessor` does
    (circuit_processor/mod.rs:130-253) — `arrays()` is what such a per-operation driver takes."""

    _OUT_KIND = {FheOp.SampleExtract: ValueKind.LWE1, FheOp.KeyswitchL1toL0: ValueKind.LWE0, FheOp.Not: ValueKind.GLWE1,
                 FheOp.GlweAdd: ValueKind.GLWE1, FheOp.CMux: ValueKind.GLWE1, FheOp.GlevCMux: ValueKind.GLEV1,
                 FheOp.MultiplyGgswGlwe: ValueKind.GLWE1, FheOp.CircuitBootstrap: ValueKind.GGSW1,
                 FheOp.SchemeSwitch: ValueKind.GGSW1, FheOp.MulXN: ValueKind.GLWE1}

    def __init__(self):
        self.op: List[int] = []        # FheOp, -1 input, -2 trivial constant
        self.kind: List[int] = []
        self.param: List[int] = []     # SampleExtract index / MulXN amount / trivial bit
        self.inputs: List[tuple] = []
        self.host: List = []           # inputs: the caller's array
        self.outputs: List[int] = []   # nodes, in add_output order

    def _add(self, op, kind, param, inputs, host=None) -> int:
        self.op.append(int(op)); self.kind.append(int(kind)); self.param.append(int(param))
        self.inputs.append(tuple(int(i) for i in inputs)); self.host.append(host)
        return len(self.op) - 1

    def add_input(self, kind: ValueKind, value: np.ndarray) -> int:
        return self._add(-1, kind, 0, (), np.ascontiguousarray(value))

    def add_trivial(self, kind: ValueKind, bit: int) -> int:
        return self._add(-2, kind, bit, ())

    def add_op(self, op: FheOp, inputs: Sequence[int], param: int = 0) -> int:
        if any(i >= len(self.op) for i in inputs):
            raise SpfError(1, "operand is not a node of this circuit")
        return self._add(FheOp(op), self._OUT_KIND[FheOp(op)], param, inputs)

    def add_output(self, node: int, kind: ValueKind) -> int:
        if self.kind[node] != int(kind):
            raise SpfError(1, "output kind does not match the node")
        self.outputs.append(int(node))
        return len(self.outputs) - 1

    def lower(self, engine: Engine):
        """-> (FheCircuit, [output arrays]) with the same nodes in the same order"""
        g = FheCircuit(engine)
        for i in range(len(self.op)):
            if self.op[i] == -1:
                n = g.add_input(ValueKind(self.kind[i]), self.host[i])
            elif self.op[i] == -2:
                n = g.add_trivial(ValueKind(self.kind[i]), self.param[i])
            else:
                n = g.add_op(FheOp(self.op[i]), self.inputs[i], self.param[i])
            assert n == i
        return g, [g.add_output(n, ValueKind(self.kind[n])) for n in self.outputs]

    def arrays(self) -> dict:
        n = len(self.op)
        cached = getattr(self, "_arrays", None)
        if cached is not None and cached[0] == (n, len(self.outputs)):
            return cached[1]
        out = self._build_arrays()
        self._arrays = ((n, len(self.outputs)), out)
        return out

    def _build_arrays(self) -> dict:
        n = len(self.op)
        ins = np.zeros((n, 3), dtype=np.uint32)
        for i, t in enumerate(self.inputs):
            ins[i, :len(t)] = t
        keep = np.zeros(n, dtype=np.uint8)
        keep[self.outputs] = 1
        return {"op": np.array([o if o >= 0 else -1 for o in self.op], dtype=np.int32), "in": ins,
                "n_in": np.array([len(t) for t in self.inputs], dtype=np.uint32),
                "param": np.array(self.param, dtype=np.uint64), "keep": keep}
