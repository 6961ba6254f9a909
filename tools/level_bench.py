"""Time of ONE level of a gate graph as a function of its width: L levels of `w` CMUX gates each, every gate of level l+1
taking two outputs of level l and one of S shared selectors (the shape of a mux_circuits block: a level tests one or a
few variables).  usage: python tools/level_bench.py [w ...]   prints microseconds per level (wall time of run() / L)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spf_amd
from spf_amd import FheCircuit, FheOp, ValueKind

widths = [int(x) for x in sys.argv[1:]] or [64, 256, 300, 512, 600, 768, 1024, 1100, 1280, 1536, 2048]
P = spf_amd.DEFAULT_128
eng = spf_amd.Engine(P, device=0)
rng = np.random.default_rng(1)
L, S = 40, 8
res = {}
for w in widths:
    g = FheCircuit(eng)
    x = [g.add_input(ValueKind.GLWE1, rng.integers(0, 1 << 64, size=P.glwe_words, dtype=np.uint64)) for _ in range(2)]
    sel = [g.add_input(ValueKind.GGSW1, (rng.standard_normal(2 * P.cbs_ggsw_complex) * 2.0 ** 40).view(np.complex128)) for _ in range(S)]
    prev = [g.add_op(FheOp.CMux, [sel[i % S], x[0], x[1]]) for i in range(w)]
    for l in range(1, L):
        prev = [g.add_op(FheOp.CMux, [sel[(i // 37 + l) % S], prev[i], prev[(i + 1) % w]]) for i in range(w)]
    g.add_output(prev[0], ValueKind.GLWE1)
    g.run()
    g.run()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        g.run()
        best = min(best, time.perf_counter() - t0)
    res[w] = round(best / L * 1e6, 1)
    print(w, res[w], "us per level", g.stats(), flush=True)
    g.close()
print(json.dumps(res))
