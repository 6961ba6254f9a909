#!/usr/bin/env python3
"""Where T blocking callers by handle spend their cycle: reads the `[pool] batch ...` lines a -DSPF_POOL_TRACE build of the library
prints (tools/ab_build.sh trace -DSPF_POOL_TRACE; SPF_HIP_LIBRARY=tools/bin/libspf_trace.so python3 tools/pool_trace_cbs.py T s
2> log) and applies Little's law per phase: callers in a phase = sum over batches of n x (time in the phase) / span.
usage: pool_trace_phases.py <log> <T>"""
import re
import sys

import numpy as np

pat = re.compile(r"\[pool\] batch op (\d+) \(by handle\) n (\d+): first member at (-?\d+) us, filled (-?\d+) us, closed->ready (-?\d+) us, "
                 r"enqueue (-?\d+) us, enqueued->event (-?\d+) us, event->marked (-?\d+) us")


def main():
    T = int(sys.argv[2])
    rows = np.array([[int(x) for x in m.groups()] for m in map(pat.search, open(sys.argv[1])) if m], dtype=np.int64)
    if not len(rows):
        raise SystemExit("no trace lines")
    t0 = rows[:, 2]
    lo, hi = np.percentile(t0, 30), np.percentile(t0, 95)
    r = rows[(t0 >= lo) & (t0 <= hi)]
    span = hi - lo
    n = r[:, 1]
    rate = n.sum() / span * 1e6
    print(f"{len(r)} batches in {span / 1e3:.0f} ms: {rate:.0f} operations/s, cycle of a caller {T / rate * 1e3:.2f} ms; batch size mean {n.mean():.0f}, "
          f"median {np.median(n):.0f}; batches of <= 16: {np.mean(n <= 16):.2f} of the batches with {n[n <= 16].sum() / n.sum():.3f} of the operations")
    names = ["first member -> closed", "closed -> ready", "ready -> enqueued", "enqueued -> event (GPU)", "event -> marked done"]
    tot = 0.0
    for i, name in enumerate(names):
        col = r[:, 3 + i].astype(float)
        # the members of a batch arrive over the filling time: half of it on average
        w = 0.5 if i == 0 else 1.0
        callers = (n * col * w).sum() / span
        tot += callers
        print(f"  {name:>26}: mean {np.average(col, weights=n):8.0f} us (weighted by size), {callers:7.1f} callers = {callers / T:.3f}")
    print(f"  {'woken -> next submit':>26}: {'':8}    {'':18} {T - tot:7.1f} callers = {(T - tot) / T:.3f}  -> {(T - tot) / rate * 1e6:.0f} us per cycle")


if __name__ == "__main__":
    main()
