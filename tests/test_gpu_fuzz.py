"""Randomised differential test of the bootstrap entry points against the oracle (r04).

Every case draws a batch size (weighted towards the sizes where the launch changes shape), an LWE dimension, the
(log_chi, log_v, body rotation) arguments, a shared or per-ciphertext LUT and the entry point (generalized bootstrap,
univariate bootstrap with the fused sample extract, circuit-bootstrap bootstrap), runs it as ONE launch through the
device-pointer forms and compares every output word with `spfo_bench_generalized_pbs`.  The default suite runs a few cases;
`SPF_FUZZ_CASES=N` (and `SPF_FUZZ_SEED`) turn it into a soak — `profiles/r04_fuzz.md` records the r04 run."""
import os

import numpy as np
import pytest

import oracle as O
import spf_amd
from util import dev_bootstrap, gpu_available, keyset, random_glwe, to_engine_params

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("SPF_FUZZ_CASES", "12"))
SEED = int(os.environ.get("SPF_FUZZ_SEED", "20260401"))
LWE_DIMS = (1, 3, 20)
FULL_N = 637   # DEFAULT_128's LWE dimension: drawn now and then with a small batch (the oracle needs 64 ms per bootstrap)
SIZES = (1, 2, 3, 7, 64, 100, 255, 256, 257, 300, 511, 512, 513, 600, 1023, 1024, 1025, 1031, 1100, 2047, 2050)


@pytest.fixture(scope="module")
def engines():
    if not gpu_available():
        pytest.skip("needs a GPU")
    out = {}
    for n in LWE_DIMS + (FULL_N,):
        ks = keyset(0x5EED0001, n, with_ksk=False)
        eng = spf_amd.Engine(to_engine_params(ks.params))
        eng.load_bootstrap_key(ks.bsk_fft)
        out[n] = (ks, eng)
    return out


def test_random_bootstrap_calls_against_the_oracle(engines):
    rng = np.random.default_rng(SEED)
    log = []
    for case in range(CASES):
        n = int(rng.choice(LWE_DIMS))
        B = int(rng.choice(SIZES)) if rng.random() < 0.8 else int(rng.integers(1, 1200))
        if rng.random() < 0.04:   # a full-length blind rotation: 637 steps, every output against the oracle
            n, B = FULL_N, int(rng.choice((1, 5, 33, 64)))
        ks, eng = engines[n]
        P = ks.params
        lwe = rng.integers(0, 1 << 64, size=(B, n + 1), dtype=np.uint64)
        if rng.random() < 0.3:   # identity steps, extreme words
            rows = rng.integers(0, B, size=max(1, B // 7))
            lwe[rows, : n] = rng.choice(np.array([0, (1 << 64) - 1, 1 << 63], dtype=np.uint64), size=(rows.size, 1))
        kind = int(rng.integers(0, 3))
        if kind == 0:      # the circuit bootstrap's bootstrap: fixed LUT, log_v = 2, body rotated by 2^62
            got = dev_bootstrap(eng, lwe)
            rot = lwe.copy()
            rot[:, -1] += np.uint64(1 << 62)
            _, exp = O.bench_generalized_pbs(rot, O.fill_cbs_lut(P), ks.bsk_fft, P, 8, 0, 2)
            what = "cbs"
        else:
            shared = rng.random() < 0.4
            lut = random_glwe(int(rng.integers(1 << 30)), 1 if shared else B, P.glwe_len)
            lut = lut[0] if shared else lut
            if kind == 1:  # univariate: log_chi = log_v = 0, sample extract fused
                got = dev_bootstrap(eng, lwe, lut, extract=True)
                _, exp = O.bench_generalized_pbs(lwe, lut, ks.bsk_fft, P, 8, 0, 0, extract=True)
                what = f"univariate lut={'shared' if shared else 'each'}"
            else:
                log_chi, log_v = int(rng.integers(0, 7)), int(rng.integers(0, 5))
                body_rotate = int(rng.integers(0, 1 << 64, dtype=np.uint64)) if rng.random() < 0.7 else 0
                got = dev_bootstrap(eng, lwe, lut, log_chi, log_v, body_rotate)
                rot = lwe.copy()
                rot[:, -1] += np.uint64(body_rotate)
                _, exp = O.bench_generalized_pbs(rot, lut, ks.bsk_fft, P, 8, log_chi, log_v)
                what = f"generalized chi={log_chi} v={log_v} rot={'random' if body_rotate else 0} lut={'shared' if shared else 'each'}"
        bad = np.nonzero((got != exp).any(axis=1))[0]
        log.append(f"{case}: n={n} B={B} {what} kernel={eng.last_blind_rotate_kernel()}")
        assert bad.size == 0, f"case {log[-1]} (seed {SEED}): {bad.size} ciphertexts differ, first {bad[:8]}"
    report = os.environ.get("SPF_FUZZ_REPORT")
    if report:
        with open(report, "w") as f:
            f.write("\n".join(log) + "\n")


@pytest.fixture(scope="module")
def tail_rig():
    if not gpu_available():
        pytest.skip("needs a GPU")
    ks = keyset(0x5EED0001, 12)
    P = ks.params
    r = O.Rng(0x7A11)
    ak, ssk = O.gen_auto_key_fft(r, ks.glwe_sk, P), O.gen_ssk_fft(r, ks.glwe_sk, P)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(ak)
    eng.load_scheme_switch_key(ssk)
    return ks, ak, ssk, eng


def test_random_tail_cmux_and_keyswitch_calls_against_the_oracle(tail_rig):
    """The rows either side of the bootstrap: keyswitch (every output), trace + rotate, scheme switch, the whole circuit
    bootstrap, CMUX with a selector per gate or shared, at random batch sizes (the trace and the scheme switch take four units
    per workgroup, the CMUX two gates per workgroup or four waves per gate); a random sample of each batch against the oracle."""
    ks, ak, ssk, eng = tail_rig
    P = ks.params
    rng = np.random.default_rng(SEED + 1)
    sizes = (1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 300, 513, 700, 1025)
    for case in range(max(4, CASES // 2)):
        B = int(rng.choice(sizes)) if rng.random() < 0.8 else int(rng.integers(1, 800))
        pick = sorted(set([0, B - 1]) | set(int(i) for i in rng.integers(0, B, size=min(B, 6))))
        kind = int(rng.integers(0, 5))
        tag = f"case {case} (seed {SEED + 1}): B={B} kind={kind}"
        if kind == 0:
            lwe1 = rng.integers(0, 1 << 64, size=(B, P.k * P.N + 1), dtype=np.uint64)
            got = eng.keyswitch_lwe_l1_lwe_l0(lwe1)
            for i in pick:
                assert np.array_equal(got[i], O.keyswitch_lwe(lwe1[i], ks.ksk, P.N, P.lwe_n, P.ks_radix_log, P.ks_count)), (tag, i)
        elif kind == 1:
            glwe = rng.integers(0, 1 << 64, size=(B, P.glwe_len), dtype=np.uint64)
            got = eng.mod_switch_trace_and_rotate(glwe)
            for i in pick:
                assert np.array_equal(got[i], O.mod_switch_trace_and_rotate(glwe[i], ak, P)), (tag, i)
        elif kind == 2:
            glev = rng.integers(0, 1 << 64, size=(B, P.cbs_count, P.glwe_len), dtype=np.uint64)
            got = eng.scheme_switch(glev)
            for i in pick:
                assert np.array_equal(got[i].view(np.float64), O.scheme_switch_fft(glev[i], ssk, P).view(np.float64)), (tag, i)
        elif kind == 3:
            lwe0 = rng.integers(0, 1 << 64, size=(B, P.lwe_n + 1), dtype=np.uint64)
            got = eng.circuit_bootstrap(lwe0)
            for i in pick[:4]:
                exp = O.circuit_bootstrap(lwe0[i], ks.bsk_fft, ak, ssk, P)
                assert np.array_equal(got[i].view(np.float64).reshape(-1), exp.view(np.float64).reshape(-1)), (tag, i)
        else:
            n_c = 2 * P.cbs_count * 2 * (P.N // 2)
            shared = rng.random() < 0.3
            g = ((rng.standard_normal((1 if shared else B, n_c)) + 1j * rng.standard_normal((1 if shared else B, n_c)))
                 * 2.0 ** 60).astype(np.complex128)
            a = rng.integers(0, 1 << 64, size=(B, P.glwe_len), dtype=np.uint64)
            b = rng.integers(0, 1 << 64, size=(B, P.glwe_len), dtype=np.uint64)
            sel = np.broadcast_to(g, (B, n_c)) if shared else g
            got = eng.cmux(np.ascontiguousarray(sel), a, b)
            for i in pick:
                assert np.array_equal(got[i], O.cmux(a[i], b[i], sel[i], P.N, P.k, P.cbs_radix_log, P.cbs_count)), (tag, i)


def test_random_gate_graphs_equal_level_by_level_evaluation(tail_rig):
    """Random gate graphs through `spf_graph_*`: levels 1 to 300 gates wide (the executor sends a level to the four-waves-per-gate
    kernel or to the streaming kernel by its width), operands drawn from every earlier level, CMux / Not / GlweAdd / MulXN mixed,
    selectors produced inside the graph (KeyswitchL1toL0 -> CircuitBootstrap, of inputs and — conversions in the middle of the
    graph — of graph values).  Every output against the same operations run level by level through the batch entry points (which
    the tests above hold against the oracle); and the same DAG pushed operation by operation through the pool by handle (pending
    results as operands: the pool's deferred scheduler) against the same expectation."""
    import tools.driver as drv
    from spf_amd import FheOp, RecordedCircuit, ValueKind
    ks, ak, ssk, eng = tail_rig
    P = ks.params
    rng = np.random.default_rng(SEED + 2)
    pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=int(rng.choice((20, 200, 2000))))
    for case in range(max(2, CASES // 6)):
        g = RecordedCircuit()                                # (lowered into a gate graph AND pushed through the pool below)
        n_in, n_sel = int(rng.integers(2, 9)), int(rng.integers(1, 7))
        glwe_in = rng.integers(0, 1 << 64, size=(n_in, P.glwe_len), dtype=np.uint64)
        lwe1_in = rng.integers(0, 1 << 64, size=(n_sel, P.k * P.N + 1), dtype=np.uint64)
        val_nodes = [g.add_input(ValueKind.GLWE1, x) for x in glwe_in]
        sel_nodes = [g.add_op(FheOp.CircuitBootstrap, [g.add_op(FheOp.KeyswitchL1toL0, [g.add_input(ValueKind.LWE1, x)])])
                     for x in lwe1_in]
        sel_vals = list(eng.circuit_bootstrap(eng.keyswitch_lwe_l1_lwe_l0(lwe1_in)))
        vals = [x for x in glwe_in]                      # eager value of val_nodes[i]
        depth = int(rng.integers(2, 8))
        for lvl in range(depth):
            if rng.random() < 0.4:                       # a conversion in the middle of the graph: a selector made of a graph value
                j, idx = int(rng.integers(0, len(val_nodes))), int(rng.integers(0, P.N))
                sel_nodes.append(g.add_op(FheOp.CircuitBootstrap, [g.add_op(FheOp.KeyswitchL1toL0, [g.add_op(FheOp.SampleExtract, [val_nodes[j]], idx)])]))
                sel_vals.append(eng.circuit_bootstrap(eng.keyswitch_lwe_l1_lwe_l0(eng.sample_extract_l1(vals[j][None], idx)))[0])
                n_sel += 1
            width = int(rng.choice((1, 2, 3, 5, 40, 70, 260, 300)))
            avail = len(val_nodes)
            ops = rng.integers(0, 4, size=width)
            ia, ib = rng.integers(0, avail, size=width), rng.integers(0, avail, size=width)
            isel = rng.integers(0, n_sel, size=width)
            amt = rng.integers(0, 4096, size=width)
            new_nodes, new_vals = [], [None] * width
            for k in range(width):
                if ops[k] == 0:
                    new_nodes.append(g.add_op(FheOp.CMux, [sel_nodes[isel[k]], val_nodes[ia[k]], val_nodes[ib[k]]]))
                elif ops[k] == 1:
                    new_nodes.append(g.add_op(FheOp.Not, [val_nodes[ia[k]]]))
                elif ops[k] == 2:
                    new_nodes.append(g.add_op(FheOp.GlweAdd, [val_nodes[ia[k]], val_nodes[ib[k]]]))
                else:
                    new_nodes.append(g.add_op(FheOp.MulXN, [val_nodes[ia[k]]], int(amt[k])))
            # the same level through the batch entry points
            A = np.stack([vals[i] for i in ia])
            Bv = np.stack([vals[i] for i in ib])
            mux = np.nonzero(ops == 0)[0]
            if mux.size:
                r = eng.cmux(np.stack([sel_vals[i] for i in isel[mux]]), A[mux], Bv[mux])
                for j, k in enumerate(mux):
                    new_vals[k] = r[j]
            for k in np.nonzero(ops == 1)[0]:
                new_vals[k] = eng.glwe_not(A[k:k + 1])[0]
            for k in np.nonzero(ops == 2)[0]:
                new_vals[k] = eng.glwe_xor(A[k:k + 1], Bv[k:k + 1])[0]
            for k in np.nonzero(ops == 3)[0]:
                new_vals[k] = eng.glwe_mul_xn(A[k:k + 1], int(amt[k]))[0]
            val_nodes += new_nodes
            vals += new_vals
        check = sorted(set(range(len(val_nodes) - len(new_nodes), len(val_nodes))) |
                       set(int(i) for i in rng.integers(n_in, len(val_nodes), size=20)))
        at = {i: g.add_output(val_nodes[i], ValueKind.GLWE1) for i in check}
        graph, outs = g.lower(eng)
        graph.run()
        for i in check:
            assert np.array_equal(outs[at[i]], vals[i]), f"graph case {case} (seed {SEED + 2}): node {i} of {len(val_nodes)}"
        graph.close()
        # the same DAG PUSHED operation by operation through the pool (pending results as operands, nothing waited for but the
        # outputs): in creation order, which is level by level here, or with every level's operations behind the next level's
        # conversions — any topological order gives the graph's words
        order = None
        if case % 2:
            order = sorted((i for i in range(len(g.op)) if g.op[i] >= 0), key=lambda i: (0 if g.op[i] in (int(FheOp.SampleExtract), int(FheOp.KeyswitchL1toL0), int(FheOp.CircuitBootstrap)) else 1, i))
            pos = {n: k for k, n in enumerate(order)}
            if any(pos.get(j, -1) > pos[i] for i in order for j in g.inputs[i] if g.op[j] >= 0):
                order = None                             # (conversions of graph values cannot all go first: creation order then)
        pushed = drv.push_circuit_by_handles(pool, g, order=order)[0]
        for i in check:
            assert np.array_equal(pushed[at[i]], vals[i]), f"pushed case {case} (seed {SEED + 2}): node {i} of {len(val_nodes)}"
    pool.close()


def test_random_parameter_sets_through_the_generic_family():
    """Random parameter sets outside the tuned kernels (N = 16 .. 1024, k = 1 .. 3, random radices and LWE dimension) and the
    N = 2048 sets with another PBS radix: a random entry point of the generic family per case — generalized / univariate /
    circuit-bootstrap PBS, CMUX, keyswitch, trace, scheme switch, whole circuit bootstrap, a small gate graph — every
    checked word against the oracle."""
    rng = np.random.default_rng(SEED + 3)
    for case in range(max(4, CASES // 2)):
        if rng.random() < 0.2:
            N, k = 2048, 1
        else:
            N, k = int(rng.choice((16, 32, 64, 128, 256, 512, 1024))), int(rng.integers(1, 4))
        if (k + 1) * N > 4096:      # keeps the oracle fast and the polynomials inside one CU's LDS
            k = 1

        def radix(max_bits=40):
            lg = int(rng.integers(1, 17))
            return lg, int(rng.integers(1, max(2, min(8, max_bits // lg) + 1)))
        pl, pc = radix()
        if N == 2048 and (pl, pc) == (16, 2):
            pl, pc = 8, 3
        cl, cc = radix(24)
        cc = min(cc, 7)
        tl, tc = radix()
        sl, sc = radix()
        kl = int(rng.integers(1, 9))
        kc = int(rng.integers(1, max(2, 32 // kl + 1)))
        n = int(rng.integers(1, 9))
        P = O.DEFAULT_128.replace(lwe_n=n, N=N, k=k, pbs_radix_log=pl, pbs_count=pc, cbs_radix_log=cl, cbs_count=cc,
                                  ks_radix_log=kl, ks_count=kc, tr_radix_log=tl, tr_count=tc, ss_radix_log=sl, ss_count=sc)
        tag = f"case {case} (seed {SEED + 3}): N={N} k={k} n={n} pbs={pc}x{pl} cbs={cc}x{cl} ks={kc}x{kl} tr={tc}x{tl} ss={sc}x{sl}"
        ks = O.gen_keyset(int(rng.integers(1 << 30)), P)
        r = O.Rng(int(rng.integers(1 << 30)))
        ak, ssk = O.gen_auto_key_fft(r, ks.glwe_sk, P), O.gen_ssk_fft(r, ks.glwe_sk, P)
        EP = to_engine_params(P).replace(tr_radix_log=tl, tr_radix_count=tc, ss_radix_log=sl, ss_radix_count=sc)
        eng = spf_amd.Engine(EP)
        eng.load_bootstrap_key(ks.bsk_fft)
        eng.load_keyswitch_key(ks.ksk)
        eng.load_automorphism_key(ak)
        eng.load_scheme_switch_key(ssk)
        B = int(rng.integers(1, 7))
        kind = int(rng.integers(0, 8))
        lwe = rng.integers(0, 1 << 64, size=(B, n + 1), dtype=np.uint64)
        if kind == 0:
            lut = random_glwe(int(rng.integers(1 << 30)), B, P.glwe_len)
            log_chi, log_v = int(rng.integers(0, 4)), int(rng.integers(0, min(4, N.bit_length() - 1)))
            got = eng.generalized_pbs(lwe, lut, log_chi, log_v, 0)
            for i in range(B):
                assert np.array_equal(got[i], O.generalized_pbs(lwe[i], lut[i], ks.bsk_fft, P, log_chi, log_v)), (tag, kind, i)
        elif kind == 1:
            lut = random_glwe(int(rng.integers(1 << 30)), 1, P.glwe_len)[0]
            got = eng.pbs_univariate(lwe, lut)
            for i in range(B):
                assert np.array_equal(got[i], O.pbs_univariate(lwe[i], lut, ks.bsk_fft, P)), (tag, kind, i)
        elif kind == 2:
            got = eng.circuit_bootstrap_pbs(lwe)
            for i in range(B):
                assert np.array_equal(got[i], O.cbs_pbs(lwe[i], ks.bsk_fft, P)), (tag, kind, i)
        elif kind == 3:
            n_c = (k + 1) * cc * (k + 1) * (N // 2)
            g = ((rng.standard_normal((B, n_c)) + 1j * rng.standard_normal((B, n_c))) * 2.0 ** 55).astype(np.complex128)
            a = rng.integers(0, 1 << 64, size=(B, P.glwe_len), dtype=np.uint64)
            b = rng.integers(0, 1 << 64, size=(B, P.glwe_len), dtype=np.uint64)
            got = eng.cmux(g, a, b)
            for i in range(B):
                assert np.array_equal(got[i], O.cmux(a[i], b[i], g[i], N, k, cl, cc)), (tag, kind, i)
        elif kind == 4:
            lwe1 = rng.integers(0, 1 << 64, size=(B, k * N + 1), dtype=np.uint64)
            got = eng.keyswitch_lwe_l1_lwe_l0(lwe1)
            for i in range(B):
                assert np.array_equal(got[i], O.keyswitch_lwe(lwe1[i], ks.ksk, k * N, n, kl, kc)), (tag, kind, i)
        elif kind == 5:
            glwe = rng.integers(0, 1 << 64, size=(B, P.glwe_len), dtype=np.uint64)
            got = eng.mod_switch_trace_and_rotate(glwe)
            glev = rng.integers(0, 1 << 64, size=(B, cc, P.glwe_len), dtype=np.uint64)
            gg = eng.scheme_switch(glev)
            for i in range(B):
                assert np.array_equal(got[i], O.mod_switch_trace_and_rotate(glwe[i], ak, P)), (tag, kind, i)
                assert np.array_equal(gg[i].view(np.float64), O.scheme_switch_fft(glev[i], ssk, P).view(np.float64)), (tag, kind, i)
        elif kind == 6:
            got = eng.circuit_bootstrap(lwe[:2])
            for i in range(min(B, 2)):
                exp = O.circuit_bootstrap(lwe[i], ks.bsk_fft, ak, ssk, P)
                assert np.array_equal(got[i].view(np.float64).reshape(-1), exp.view(np.float64).reshape(-1)), (tag, kind, i)
        else:
            # a gate graph at this parameter set: KeyswitchL1toL0 -> CircuitBootstrap -> a chain of CMux gates over scattered operands
            from spf_amd import FheOp, ValueKind
            nb = min(B, 2)
            lwe1 = rng.integers(0, 1 << 64, size=(nb, k * N + 1), dtype=np.uint64)
            a = rng.integers(0, 1 << 64, size=(nb + 1, P.glwe_len), dtype=np.uint64)
            g = spf_amd.FheCircuit(eng)
            sel = [g.add_op(FheOp.CircuitBootstrap, [g.add_op(FheOp.KeyswitchL1toL0, [g.add_input(ValueKind.LWE1, x)])]) for x in lwe1]
            ai = [g.add_input(ValueKind.GLWE1, x) for x in a]
            acc = ai[nb]
            for i in range(nb):
                acc = g.add_op(FheOp.CMux, [sel[i], g.add_op(FheOp.Not, [acc]), ai[i]])
            out = g.add_output(acc, ValueKind.GLWE1)
            g.run()
            exp = a[nb]
            for i in range(nb):
                l0 = O.keyswitch_lwe(lwe1[i], ks.ksk, k * N, n, kl, kc)
                exp = O.cmux(O.glwe_not(exp, N, k), a[i], O.circuit_bootstrap(l0, ks.bsk_fft, ak, ssk, P), N, k, cl, cc)
            assert np.array_equal(out, exp), (tag, kind)
            g.close()
        eng.close()
