#!/usr/bin/env python3
"""How many CMux gates of a topological level of the 32 x 32 multiplier graph (BASELINE config 5) select on the SAME GGSW?
Builds the gate graph of `jobs` multiplications with a recording stand-in for FheCircuit (no GPU) and prints, per level,
gates / distinct selectors.  usage: python tools/graph_selector_sharing.py [jobs]"""
import collections
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spf_amd.graph import FheOp, ValueKind  # noqa: E402
from spf_amd.mux_circuits import GraphBuilder, append_uint_multiply, parse_mux_circuit  # noqa: E402


class Recorder:
    def __init__(self):
        self.level, self.op, self.ins = [], [], []

    def _add(self, op, ins, level):
        self.level.append(level); self.op.append(op); self.ins.append(list(ins))
        return len(self.level) - 1

    def add_input(self, kind, value=None):
        return self._add("in", [], 0)

    def add_trivial(self, kind, bit):
        return self._add("triv", [], 0)

    def add_op(self, op, inputs, param=0):
        return self._add(op, inputs, 1 + max(self.level[i] for i in inputs))


def main():
    jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    blk16 = parse_mux_circuit(open(os.path.join(ROOT, "spf_amd", "data", "mux_multiplier_n16_m16.bincode"), "rb").read())
    g = Recorder()
    b = GraphBuilder(g)
    for _ in range(jobs):
        sel = [b.to_ggsw(g.add_input(ValueKind.GLWE1)) for _ in range(64)]
        append_uint_multiply(b, sel[:32], sel[32:], lambda x, y: {(16, 16): blk16}[(x, y)])
    by_level = collections.defaultdict(list)
    for i, (o, l) in enumerate(zip(g.op, g.level)):
        if o == FheOp.CMux:
            by_level[l].append(i)
    rows = []
    for l in sorted(by_level):
        sels = collections.Counter(g.ins[i][0] for i in by_level[l])
        rows.append((l, len(by_level[l]), len(sels), max(sels.values())))
    gates, distinct = sum(r[1] for r in rows), sum(r[2] for r in rows)
    print(f"{jobs} job(s): {len(rows)} levels with CMux, {gates} gates, {gates / distinct:.1f} gates per distinct selector and level")
    print(f"level width median {statistics.median(r[1] for r in rows)}, distinct selectors per level median {statistics.median(r[2] for r in rows)}")
    print("share of gates in levels with >= 4 gates per selector:", round(sum(r[1] for r in rows if r[1] / r[2] >= 4) / gates, 4))
    print("(level, gates, distinct selectors, largest run) every 40th level:")
    for r in rows[::40]:
        print("  ", r)


if __name__ == "__main__":
    main()
