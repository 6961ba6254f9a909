"""Gate pool over several GPUs (BASELINE config 5: "32x32-bit encrypted multiply via mux_circuits ... 8xMI355X gate
pool"; reference shape: one `CircuitProcessor` per machine fed whole `FheCircuit`s,
parasol_runtime/src/circuit_processor/mod.rs:573-623, multiplier circuits from
parasol_runtime/src/circuits/mul.rs:90-200).

A gate graph is a chain of dependent CMUX levels (a multiplier block is 126 levels deep) hanging off one wide
level of independent conversions (SampleExtract -> KeyswitchL1toL0 -> CircuitBootstrap, one per input bit).
Splitting ONE graph across GPUs would put a 256 KiB GGSW per selector on the wire for every gate that crosses
the cut; independent graphs need nothing.  So the unit of sharding is the JOB (one circuit evaluation): jobs
are dealt to ranks by cost (longest-processing-time first), every rank lowers ITS jobs into one
`spf_amd.FheCircuit` — so the level-batching executor still sees wide levels: K jobs x gates per level — runs
it on its own GPU with its own key replica, and the small results (32 KiB per output bit) are gathered.  No
data-path collective; the only exchange is the result gather (torch.distributed, "nccl" = RCCL over xGMI, or
gloo in the CPU tests).

Everything here is host-side plumbing: the executor is passed in, so the CPU tests drive it with the plaintext
evaluator of `spf_amd.mux_circuits` under gloo, and `bench.py` with the GPU engine.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


def lpt_shards(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of jobs to ranks; deterministic (ties by job index), every rank
    computes the same table.  Returns, per rank, its job indices in ascending order."""
    if world <= 0:
        raise ValueError("world must be positive")
    load = [0.0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for j in sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i)):
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(j)
        load[r] += float(costs[j])
    return [sorted(s) for s in shards]


def run_sharded(jobs: Sequence, costs: Sequence[float], rank: int, world: int,
                run_batch: Callable[[List], List[np.ndarray]], dist=None, dst: int = 0) -> Optional[List[np.ndarray]]:
    """Run `jobs` over `world` ranks: this rank executes its shard with `run_batch(list of jobs) -> list of result
    arrays` (ONE call, so the executor can lower them into one graph) and the results are gathered in job order on
    rank `dst` (None elsewhere).  `dist` is torch.distributed (already initialised) when world > 1."""
    if len(jobs) != len(costs):
        raise ValueError("one cost per job")
    shard = lpt_shards(costs, world)[rank]
    local = run_batch([jobs[i] for i in shard]) if shard else []
    if len(local) != len(shard):
        raise RuntimeError("run_batch must return one result per job")
    mine: Dict[int, np.ndarray] = {i: np.ascontiguousarray(r) for i, r in zip(shard, local)}
    if world == 1:
        return [mine[i] for i in range(len(jobs))]
    gathered = [None] * world if rank == dst else None
    dist.gather_object(mine, gathered, dst=dst)
    if rank != dst:
        return None
    merged: Dict[int, np.ndarray] = {}
    for part in gathered:
        merged.update(part)
    if sorted(merged) != list(range(len(jobs))):
        raise RuntimeError("gate pool: a job result is missing or duplicated")
    return [merged[i] for i in range(len(jobs))]


def circuit_jobs_as_one_graph(engine, circuit, bit_glwes: np.ndarray, record: bool = False):
    """Lower K evaluations of a `MuxCircuit` into ONE gate graph the way `add_circuit` / `mul_impl` feed their
    blocks (`FheCircuit::insert_mux_circuit_and_connect_inputs`, fhe_circuit.rs:473-494; circuits/add.rs:10-32,
    circuits/mul.rs:104-117): per input bit an L1 GLWE -> SampleExtract(0) -> KeyswitchL1toL0 -> CircuitBootstrap,
    the GGSWs select the block's CMUX tree.
    bit_glwes: [K, n_inputs, glwe_words] uint64.  Returns (graph, outs) with outs[k][o] the array that
    graph.run() fills with output bit o of evaluation k.  record=True: the graph is a `RecordedCircuit` (not bound to
    an executor) and outs[k][o] is the index of the output in its `outputs` list."""
    from .graph import FheCircuit, FheOp, RecordedCircuit, ValueKind
    from .mux_circuits import insert_mux_circuit
    K, n_in = bit_glwes.shape[0], bit_glwes.shape[1]
    if n_in != circuit.n_inputs:
        raise ValueError("one GLWE per circuit input")
    g = RecordedCircuit() if record else FheCircuit(engine)
    outs = []
    for k in range(K):
        sel = []
        for i in range(n_in):
            x = g.add_input(ValueKind.GLWE1, bit_glwes[k, i])
            x = g.add_op(FheOp.SampleExtract, [x], 0)
            x = g.add_op(FheOp.KeyswitchL1toL0, [x])
            sel.append(g.add_op(FheOp.CircuitBootstrap, [x]))
        outs.append([g.add_output(n, ValueKind.GLWE1) for n in insert_mux_circuit(g, circuit, sel)])
    return g, outs


multiply_jobs_as_one_graph = circuit_jobs_as_one_graph
