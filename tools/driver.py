"""The native caller drivers of tests and bench.py (tools/pool_driver.cpp), built on demand beside the library: load() returns
the ctypes handle with prototypes; run_circuit_by_handles() walks a `spf_amd.RecordedCircuit` node by node through the pool's
submits by handle the way the reference's `CircuitProcessor` walks an `FheCircuit` (circuit_processor/mod.rs:130-253).
Test infrastructure: uses the public C ABI only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DRV = None


def load():
    global _DRV
    if _DRV is not None:
        return _DRV
    drv_path = os.path.join(ROOT, "tools", "bin", "libpool_driver.so")
    src = os.path.join(ROOT, "tools", "pool_driver.cpp")
    hdr = os.path.join(ROOT, "include", "spf_hip.h")
    if not os.path.exists(drv_path) or os.path.getmtime(drv_path) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(drv_path), exist_ok=True)
        tmp = f"{drv_path}.{os.getpid()}.tmp"
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-I", os.path.join(ROOT, "include"),
                        "-o", tmp, src], check=True)
        os.replace(tmp, drv_path)
    d = C.CDLL(drv_path)
    P, D = C.c_void_p, C.POINTER(C.c_double)
    d.spf_pool_drive.restype = C.c_long
    d.spf_pool_drive.argtypes = [P, P, P, C.c_int, C.c_double, P, C.c_size_t, C.c_size_t, D, D]
    d.spf_pool_drive_collect.restype = C.c_long
    d.spf_pool_drive_collect.argtypes = [P, P, P, C.c_int, C.c_double, P, C.c_size_t, C.c_size_t, D, P]
    d.spf_pool_drive_v.restype = C.c_long
    d.spf_pool_drive_v.argtypes = [P, P, P, P, C.c_int, C.c_double, P, D, P]
    d.spf_pool_drive_cmux.restype = C.c_long
    d.spf_pool_drive_cmux.argtypes = [P, P, P, C.c_int, C.c_int, C.c_double, P, C.c_size_t, P, P, C.c_size_t, D]
    d.spf_pool_drive_cmux_v.restype = C.c_long
    d.spf_pool_drive_cmux_v.argtypes = [P, P, P, P, C.c_int, C.c_int, C.c_double, P, P, P, D]
    d.spf_pool_push_cmux_v.restype = C.c_long
    d.spf_pool_push_cmux_v.argtypes = [P, P, P, P, C.c_int, C.c_int, C.c_double, P, P, P, D]
    d.spf_circuit_push.restype = C.c_int
    d.spf_circuit_push.argtypes = [P, P, P, P, P, C.c_uint32, P, P, P, P, P, P, P, C.c_uint32, P, C.c_uint32, D]
    d.spf_circuit_drive.restype = C.c_int
    d.spf_circuit_drive.argtypes = [P, P, P, P, C.c_int, C.c_uint32, P, P, P, P, P, P, D]
    _DRV = d
    return d


def fn(lib, name):
    return C.cast(getattr(lib, name), C.c_void_p)


def handles(values):
    """a C array of spf_value* from spf_amd.Value objects"""
    return (C.c_void_p * len(values))(*[v._h for v in values])


def upload_circuit_inputs(pool, rec, member=-1):
    """values[i] for the input / constant nodes of a RecordedCircuit (None elsewhere)"""
    vals = [None] * len(rec.op)
    by_kind = {}
    for i, op in enumerate(rec.op):
        if op == -1:
            by_kind.setdefault(rec.kind[i], []).append(i)
        elif op == -2:
            vals[i] = pool.trivial(rec.kind[i], rec.param[i], member)
    for kind, nodes in by_kind.items():   # the inputs of one kind go up together (spf_value_upload_batch: one block, one copy)
        up = pool.upload_batch(kind, np.stack([np.ascontiguousarray(rec.host[i]).view(np.uint64).reshape(-1) for i in nodes]), member)
        for i, v in zip(nodes, up):
            vals[i] = v
    return vals


def run_circuit_by_handles(pool, rec, threads=64, member=-1, vals=None):
    """-> (outputs as arrays in rec.outputs order, seconds inside the driver, seconds of upload + driver + download).  Every
    operation of the circuit is ONE spf_pool_submit_op_v + spf_pool_wait from one of `threads` native workers.  The second
    figure is what spf_graph_run's time stands against: the inputs go up (one copy per kind), the DAG is walked, the outputs come
    back (one gathered copy); the Python bookkeeping around the three calls (handle tables, wrappers) is not in it."""
    import time
    from spf_amd import Value
    d = load()
    lib = pool._lib
    a = rec.arrays()
    n = len(rec.op)
    t_up = 0.0
    if vals is None:
        t0 = time.perf_counter()
        vals = upload_circuit_inputs(pool, rec, member)
        t_up = time.perf_counter() - t0
    table = (C.c_void_p * n)(*[(v._h if v is not None else None) for v in vals])
    el = C.c_double()
    st = d.spf_circuit_drive(pool._h, fn(lib, "spf_pool_submit_op_v"), fn(lib, "spf_pool_wait"), fn(lib, "spf_value_release"),
                             threads, n, a["op"].ctypes.data, a["in"].ctypes.data, a["n_in"].ctypes.data,
                             a["param"].ctypes.data, table, a["keep"].ctypes.data, C.byref(el))
    # the driver released (and nulled) what nobody keeps; inputs it released must not be released again by their wrappers
    for i, v in enumerate(vals):
        if v is not None and not table[i]:
            v._h = None
    kept = {}
    for node in rec.outputs:
        if node not in kept:
            if not table[node]:
                kept[node] = None
            elif vals[node] is not None:
                kept[node] = vals[node]
            else:
                kept[node] = Value(pool, C.c_void_p(table[node]))
    if st != 0:
        for v in kept.values():
            if v is not None:
                v.release()
        raise RuntimeError(f"spf_circuit_drive: status {st}")
    order = [kept[node] for node in rec.outputs]
    kinds = {rec.kind[node] for node in rec.outputs}
    t0 = time.perf_counter()
    if len(kinds) == 1 and len(order) > 1:   # (one gathered copy, one call)
        outs = list(pool.download_batch(order))
    else:
        outs = [v.download() for v in order]
    t_down = time.perf_counter() - t0
    for v in kept.values():
        v.release()
    for v in vals:
        if v is not None:
            v.release()
    return outs, el.value, t_up + el.value + t_down


def push_circuit_by_handles(pool, rec, member=-1, order=None, flush_conversions=True):
    """-> (outputs as arrays in rec.outputs order, seconds inside the pusher, seconds of upload + pusher + download).  ONE native
    thread submits every operation of the circuit without a ticket and without a wait, level by level (operands that are still
    pending: include/spf_hip.h "Deferred operands"), then waits for the output values only."""
    import time
    import numpy as np
    from spf_amd import Value
    d = load()
    lib = pool._lib
    a = rec.arrays()
    n = len(rec.op)
    level = np.zeros(n, dtype=np.int64)
    for i in range(n):
        if rec.op[i] >= 0:
            level[i] = 1 + max((level[j] for j in rec.inputs[i]), default=0)
    if order is None:   # level by level; any topological order of the operation nodes will do (the pool batches by depth)
        order = [i for i in np.argsort(level, kind="stable") if rec.op[i] >= 0]
    order = np.array(order, dtype=np.uint32)
    outputs = np.array(rec.outputs, dtype=np.uint32)
    t0 = time.perf_counter()
    vals = upload_circuit_inputs(pool, rec, member)
    t_up = time.perf_counter() - t0
    table = (C.c_void_p * n)(*[(v._h if v is not None else None) for v in vals])
    el = C.c_double()
    st = d.spf_circuit_push(pool._h, fn(lib, "spf_pool_submit_op_v"), fn(lib, "spf_value_wait"), fn(lib, "spf_value_release"),
                            fn(lib, "spf_pool_flush") if flush_conversions else None, n,
                            a["op"].ctypes.data, a["in"].ctypes.data, a["n_in"].ctypes.data, a["param"].ctypes.data, table,
                            a["keep"].ctypes.data, order.ctypes.data, len(order), outputs.ctypes.data, len(outputs), C.byref(el))
    for i, v in enumerate(vals):   # inputs the pusher released must not be released again by their wrappers
        if v is not None and not table[i]:
            v._h = None
    kept = {}
    for node in rec.outputs:
        if node not in kept:
            kept[node] = None if not table[node] else (vals[node] if vals[node] is not None else Value(pool, C.c_void_p(table[node])))
    if st != 0:
        for v in kept.values():
            if v is not None:
                v.release()
        raise RuntimeError(f"spf_circuit_push: status {st}")
    order_v = [kept[node] for node in rec.outputs]
    kinds = {rec.kind[node] for node in rec.outputs}
    t0 = time.perf_counter()
    if len(kinds) == 1 and len(order_v) > 1:
        outs = list(pool.download_batch(order_v))
    else:
        outs = [v.download() for v in order_v]
    t_down = time.perf_counter() - t0
    for v in kept.values():
        v.release()
    for v in vals:
        if v is not None:
            v.release()
    return outs, el.value, t_up + el.value + t_down
