"""Device-resident values across the per-operation boundary (include/spf_hip.h "device-resident values"; VERDICT r05 task 1).

The reference's `CircuitProcessor::exec_op` calls `Evaluation` once per `FheOp` (circuit_processor/mod.rs:255-540); the GGSW
a `CircuitBootstrap` makes is consumed by the CMux gates behind it (fhe_circuit.rs:473-494).  By handle nothing crosses PCIe
between the two.  Checked here: every operation kind by handle equals the oracle / the batch entry points word for word; the
reference's 32-bit adder driven node by node from 64 native threads through the pool by handles decrypts to a + b and is
word-equal to `spf_graph_run` of the same circuit; misuse is an error, never a read of unfinished data; released values give
their memory back (hipMemGetInfo returns to the baseline)."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle as O
import spf_amd
from spf_amd import FheOp, ValueKind
from tests.util import keyset, random_glwe, random_lwe_batch, to_engine_params

pytestmark = pytest.mark.gpu

SMALL_N = 12


@pytest.fixture(scope="module")
def rig():
    ks = keyset(0x5EED0001, SMALL_N)
    P = ks.params
    r = O.Rng(0x7A11)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(O.gen_auto_key_fft(r, ks.glwe_sk, P))
    ssk = O.gen_ssk_fft(r, ks.glwe_sk, P)
    eng.load_scheme_switch_key(ssk)
    return ks, eng, ssk


def test_every_operation_kind_by_handle_against_the_oracle(rig):
    """All eleven pool operations by handle from many threads, one ciphertext per call, chained the way a circuit chains them
    (the GGSW of KeyswitchL1toL0 -> CircuitBootstrap selects the CMux that follows, without leaving HBM)."""
    ks, eng, ssk = rig
    P, EP = ks.params, eng.params
    n = 12
    r = np.random.default_rng(191)
    lwe1 = random_lwe_batch(190, n, P.N * P.k)
    a = random_glwe(192, n, P.glwe_len)
    b = random_glwe(193, n, P.glwe_len)
    ga = random_glwe(194, n * P.cbs_count, P.glwe_len).reshape(n, P.cbs_count * P.glwe_len)
    gb = random_glwe(195, n * P.cbs_count, P.glwe_len).reshape(n, P.cbs_count * P.glwe_len)
    ggsw = ((r.standard_normal((n, EP.cbs_ggsw_complex)) + 1j * r.standard_normal((n, EP.cbs_ggsw_complex))) * 2.0 ** 58)
    pool = spf_amd.Pool(eng, max_batch=64, max_wait_us=2000)
    got = {}

    def task(i):
        va, vb = pool.upload(ValueKind.GLWE1, a[i]), pool.upload(ValueKind.GLWE1, b[i])
        vga, vgb = pool.upload(ValueKind.GLEV1, ga[i]), pool.upload(ValueKind.GLEV1, gb[i])
        vg = pool.upload(ValueKind.GGSW1, ggsw[i])
        vl = pool.upload(ValueKind.LWE1, lwe1[i])
        l0 = pool.run_v(FheOp.KeyswitchL1toL0, [vl])
        sel = pool.run_v(FheOp.CircuitBootstrap, [l0])
        sel2 = pool.keyswitch_circuit_bootstrap_v(vl)
        res = {
            "l0": l0.download(), "sel": sel.download(), "sel2": sel2.download(),
            "cmux": pool.run_v(FheOp.CMux, [sel, va, vb]).download(),
            "se": pool.run_v(FheOp.SampleExtract, [va], 0 if i % 2 else 1234).download(),
            "not": pool.run_v(FheOp.Not, [va]).download(),
            "add": pool.run_v(FheOp.GlweAdd, [va, vb]).download(),
            "xn": pool.run_v(FheOp.MulXN, [va], 4096 + 77).download(),
            "mul": pool.run_v(FheOp.MultiplyGgswGlwe, [vg, va]).download(),
            "gc": pool.run_v(FheOp.GlevCMux, [vg, vga, vgb]).download(),
            "ss": pool.run_v(FheOp.SchemeSwitch, [vga]).download(),
        }
        got[i] = res
        return i

    try:
        with ThreadPoolExecutor(max_workers=n) as ex:
            assert sorted(ex.map(task, range(n))) == list(range(n))
        c = pool.counters()
        assert c["handle_ops"] == 11 * n and c["handle_launches"] < c["handle_ops"], c   # coalesced
        with pytest.raises(spf_amd.SpfError):
            pool.run_v(FheOp.SampleExtract, [pool.upload(ValueKind.GLWE1, a[0])], P.N)   # index out of range: refused at submit
    finally:
        import gc
        gc.collect()
        pool.close()
    exp_l0 = eng.keyswitch_lwe_l1_lwe_l0(lwe1)
    exp_sel = eng.circuit_bootstrap(exp_l0)
    exp_cmux = eng.cmux(exp_sel, a, b)
    for i in range(n):
        g = got[i]
        assert np.array_equal(g["l0"], O.keyswitch_lwe(lwe1[i], ks.ksk, P.N, SMALL_N, P.ks_radix_log, P.ks_count)), i
        assert np.array_equal(g["l0"], exp_l0[i]), i
        assert np.array_equal(g["sel"].view(np.float64), exp_sel[i].view(np.float64)), i
        assert np.array_equal(g["sel2"].view(np.float64), exp_sel[i].view(np.float64)), i
        assert np.array_equal(g["cmux"], exp_cmux[i]), i
        assert np.array_equal(g["cmux"], O.cmux(a[i], b[i], exp_sel[i], P.N, P.k, P.cbs_radix_log, P.cbs_count)), i
        assert np.array_equal(g["se"], O.sample_extract(a[i], 0 if i % 2 else 1234, P.N, P.k)), i
        assert np.array_equal(g["not"], O.glwe_not(a[i], P.N, P.k)), i
        assert np.array_equal(g["add"], O.glwe_xor(a[i], b[i], P.N, P.k)), i
        assert np.array_equal(g["xn"], O.glwe_mul_xn(a[i], 77, P.N, P.k)), i
        fft = O.glwe_ggsw_mad(np.zeros(P.glwe_len // 2, dtype=np.complex128), a[i], ggsw[i], P.N, P.k, P.cbs_radix_log, P.cbs_count)
        assert np.array_equal(g["mul"], np.concatenate([O.poly_ifft(fft[:P.N // 2]), O.poly_ifft(fft[P.N // 2:])])), i
        for j in range(P.cbs_count):
            exp = O.cmux(ga[i].reshape(P.cbs_count, -1)[j], gb[i].reshape(P.cbs_count, -1)[j], ggsw[i], P.N, P.k, P.cbs_radix_log, P.cbs_count)
            assert np.array_equal(g["gc"].reshape(P.cbs_count, P.glwe_len)[j], exp), (i, j)
        assert np.array_equal(g["ss"].view(np.float64), O.scheme_switch_fft(ga[i].reshape(P.cbs_count, -1), ssk, P).view(np.float64)), i


def test_misuse_is_an_error_not_a_read_of_unfinished_data(rig):
    ks, eng, _ = rig
    P = ks.params
    pool = spf_amd.Pool(eng, max_batch=64, max_wait_us=200000)   # (a long quiet time: the batch below stays open)
    other = spf_amd.Pool(eng, max_batch=64, max_wait_us=200)
    try:
        a = pool.upload(ValueKind.GLWE1, random_glwe(7, 1, P.glwe_len)[0])
        lwe1 = pool.upload(ValueKind.LWE1, random_lwe_batch(8, 1, P.N * P.k)[0])
        assert a.info() == {"kind": 2, "bytes": P.glwe_len * 8, "member": 0, "valid": True}
        with pytest.raises(spf_amd.SpfError):   # wrong kind in an operand slot
            pool.submit_v(FheOp.CMux, [a, a, a])
        with pytest.raises(spf_amd.SpfError):   # wrong arity
            pool.submit_v(FheOp.GlweAdd, [a])
        with pytest.raises(spf_amd.SpfError):   # wrong size at upload
            pool.upload(ValueKind.GLWE1, np.zeros(5, dtype=np.uint64))
        with pytest.raises(spf_amd.SpfError):   # a value of another pool
            other.submit_v(FheOp.Not, [a])
        pending, ticket = pool.submit_v(FheOp.KeyswitchL1toL0, [lwe1])
        assert pending.info()["valid"] is False
        with pytest.raises(spf_amd.SpfError):   # not run yet: reading it is refused, not a read of unfinished data
            pending.download()
        with pytest.raises(spf_amd.SpfError):
            pending.device_ptr()
        with pytest.raises(spf_amd.SpfError):   # ... and another pool will not order itself behind this one
            other.submit_v(FheOp.CircuitBootstrap, [pending])
        lwe1.release()                          # an operand may go as soon as the submit has returned
        pool.wait(ticket)
        assert pending.info()["valid"] is True
        assert np.array_equal(pending.download(), eng.keyswitch_lwe_l1_lwe_l0(random_lwe_batch(8, 1, P.N * P.k))[0])
        # constants: the trivial encryptions of a graph (spf_graph_add_trivial) as values
        one = pool.trivial(ValueKind.GLWE1, 1).download()
        assert one[P.N * P.k] == 1 << 63 and np.count_nonzero(one) == 1
        glev = pool.trivial(ValueKind.GLEV1, 1).download().reshape(P.cbs_count, -1)
        assert [int(glev[j, P.N * P.k]) for j in range(P.cbs_count)] == [1 << (64 - P.cbs_radix_log * (j + 1)) for j in range(P.cbs_count)]
        for bit in (0, 1):
            assert np.array_equal(pool.trivial(ValueKind.GGSW1, bit).download().view(np.float64),
                                  eng.l1ggsw_constant(bit).view(np.float64))
    finally:
        import gc
        gc.collect()
        pool.close()
        other.close()


def test_pending_results_as_operands_a_chain_pushed_without_a_wait(rig):
    """Deferred operands (include/spf_hip.h): a result that is still pending is an operand of the next submit; the pool orders the
    batches on the device and launches what was pushed when somebody waits for a result.  A whole chain — SampleExtract ->
    KeyswitchL1toL0 -> CircuitBootstrap -> CMux(sel, Not(a), GlweAdd(a, b)) -> CMux again — is pushed from ONE thread for twelve
    independent inputs without a single wait and without tickets, then only the last values are waited for: every word equals
    the operation-by-operation entry points, the pushed operations of a kind and level ran as ONE launch each, and a chain on a
    failing producer fails as a whole."""
    ks, eng, _ = rig
    P = ks.params
    n = 12
    a, b = random_glwe(301, n, P.glwe_len), random_glwe(302, n, P.glwe_len)
    x = random_glwe(303, n, P.glwe_len)
    pool = spf_amd.Pool(eng, max_batch=64, max_wait_us=100000)   # (a long quiet time: only the waits below launch anything)
    try:
        va, vb, vx = pool.upload_batch(ValueKind.GLWE1, a), pool.upload_batch(ValueKind.GLWE1, b), pool.upload_batch(ValueKind.GLWE1, x)
        c0 = pool.counters()
        last, mid = [], []
        for i in range(n):
            se = pool.push_v(FheOp.SampleExtract, [vx[i]], 3)          # operands valid: an ordinary batch, still open
            l0 = pool.push_v(FheOp.KeyswitchL1toL0, [se])              # pending operand: deferred, depth 1
            sel = pool.push_v(FheOp.CircuitBootstrap, [l0])            # depth 2
            nt = pool.push_v(FheOp.Not, [va[i]])
            ad = pool.push_v(FheOp.GlweAdd, [va[i], vb[i]])
            m1 = pool.push_v(FheOp.CMux, [sel, nt, ad])                # depth 3
            m2 = pool.push_v(FheOp.CMux, [sel, m1, pool.push_v(FheOp.MulXN, [m1], 5)])   # depth 5 (MulXN at 4)
            assert not m2.info()["valid"]
            mid.append(m1)
            last.append(m2)
        got2 = [v.wait().download() for v in last]                     # the first wait launches everything pushed so far
        got1 = [v.download() for v in mid]                             # valid: they ran before their users
        c1 = pool.counters()
        assert c1["handle_ops"] - c0["handle_ops"] == 8 * n
        assert c1["handle_launches"] - c0["handle_launches"] == 8, c1   # one launch per kind and level
        # a ticket for a deferred operation works like any other ticket
        v, t = pool.submit_v(FheOp.Not, [pool.push_v(FheOp.Not, [va[0]])])
        pool.wait(t)
        assert np.array_equal(v.download(), a[0])
    finally:
        import gc
        gc.collect()
        pool.close()
    e_sel = eng.circuit_bootstrap(eng.keyswitch_lwe_l1_lwe_l0(eng.sample_extract_l1(x, 3)))
    e_m1 = eng.cmux(e_sel, eng.glwe_not(a), eng.glwe_xor(a, b))
    e_m2 = eng.cmux(e_sel, e_m1, eng.glwe_mul_xn(e_m1, 5))
    for i in range(n):
        assert np.array_equal(got1[i], e_m1[i]), i
        assert np.array_equal(got2[i], e_m2[i]), i
        assert np.array_equal(got1[i], O.cmux(O.glwe_not(a[i], P.N, P.k), O.glwe_xor(a[i], b[i], P.N, P.k), e_sel[i], P.N, P.k, P.cbs_radix_log, P.cbs_count)), i

    # a pool that is closed with a pushed chain nobody has waited for: spf_pool_destroy launches and drains it, the results are
    # valid afterwards (a value may outlive its pool)
    pool = spf_amd.Pool(eng, max_batch=64, max_wait_us=100000)
    va0, vx0 = pool.upload(ValueKind.GLWE1, a[0]), pool.upload(ValueKind.GLWE1, x[0])
    sel = pool.push_v(FheOp.CircuitBootstrap, [pool.push_v(FheOp.KeyswitchL1toL0, [pool.push_v(FheOp.SampleExtract, [vx0], 3)])])
    left = pool.push_v(FheOp.CMux, [sel, va0, pool.push_v(FheOp.Not, [va0])])
    assert not left.info()["valid"]
    pool.close()
    assert left.info()["valid"] and left.wait() is left
    assert np.array_equal(left.download(), eng.cmux(e_sel[0:1], a[0:1], eng.glwe_not(a[0:1]))[0])
    del left, sel, va0, vx0

    # a producer that fails (no keys in this context): everything pushed behind it fails with it, nothing is read, the pool goes on
    bare = spf_amd.Engine(to_engine_params(P))
    pool = spf_amd.Pool(bare, max_batch=16, max_wait_us=100000)
    try:
        g = pool.upload(ValueKind.GLWE1, a[0])
        l0 = pool.push_v(FheOp.KeyswitchL1toL0, [pool.push_v(FheOp.SampleExtract, [g], 0)])   # no keyswitch key: this batch fails
        sel = pool.push_v(FheOp.CircuitBootstrap, [l0])
        out = pool.push_v(FheOp.CMux, [sel, g, g])
        with pytest.raises(spf_amd.SpfError):
            out.wait()
        assert not out.info()["valid"] and not sel.info()["valid"] and not l0.info()["valid"]
        with pytest.raises(spf_amd.SpfError):
            pool.push_v(FheOp.Not, [out])                                # an operand whose producer failed is refused
        assert np.array_equal(pool.push_v(FheOp.Not, [g]).wait().download(), O.glwe_not(a[0], P.N, P.k))
    finally:
        import gc
        gc.collect()
        pool.close()
        bare.close()


def test_reference_adder_pushed_from_one_thread_without_waits():
    """`mux_circuits::add::ripple_carry_adder(32, 32, false)`: all 1 871 operations pushed by ONE native thread in the order the
    reference's processor would make them ready (level by level), no ticket, no wait; then the 33 outputs are waited for
    (tools/pool_driver.cpp: spf_circuit_push).  Word-equal to spf_graph_run of the same circuit, and the pool made about one
    launch per kind and level; afterwards no value is left alive."""
    import tools.driver as drv
    from spf_amd.gate_pool import circuit_jobs_as_one_graph
    from spf_amd.mux_circuits import ripple_carry_adder
    ks = keyset(0x5EED0002, SMALL_N)
    P = ks.params
    r = O.Rng(0x7A12)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(O.gen_auto_key_fft(r, ks.glwe_sk, P))
    eng.load_scheme_switch_key(O.gen_ssk_fft(r, ks.glwe_sk, P))
    A, B = 0x9E3779B9, 0x7F4A7C15
    cts = _adder_inputs(ks, r, A, B)[None]
    rec, _ = circuit_jobs_as_one_graph(eng, ripple_carry_adder(32, 32, False), cts, record=True)
    g, g_outs = rec.lower(eng)
    g.run()
    st = g.stats()
    pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=50)
    try:
        c0 = pool.counters()
        outs, inner, whole = drv.push_circuit_by_handles(pool, rec)
        c1 = pool.counters()
        assert len(outs) == 33 and all(np.array_equal(x, y) for x, y in zip(outs, g_outs))
        total = 0
        for i, o in enumerate(outs):
            total |= int(O.decode(O.decrypt_glwe_raw(o, ks.glwe_sk, P.N, P.k)[0], 1)) << i
        assert total == A + B
        n_ops = c1["handle_ops"] - c0["handle_ops"]
        n_l = c1["handle_launches"] - c0["handle_launches"]
        assert n_ops == len([o for o in rec.op if o >= 0])
        assert n_l <= 3 * st["launches"], (n_l, st)     # level batches, not one launch per operation
        # any topological order will do: the nodes in creation order (depth first through every gate's sub-circuit, not level
        # by level), and a random topological order — the pool sorts what it is given into (depth, kind) batches
        creation = [i for i in range(len(rec.op)) if rec.op[i] >= 0]
        outs, _, _ = drv.push_circuit_by_handles(pool, rec, order=creation)
        assert all(np.array_equal(x, y) for x, y in zip(outs, g_outs))
        rng = np.random.default_rng(5)
        prio = rng.random(len(rec.op))
        import heapq
        n_wait = [sum(1 for j in rec.inputs[i] if rec.op[j] >= 0) if rec.op[i] >= 0 else 0 for i in range(len(rec.op))]
        users = [[] for _ in rec.op]
        for i in creation:
            for j in rec.inputs[i]:
                if rec.op[j] >= 0:
                    users[j].append(i)
        heap = [(prio[i], i) for i in creation if n_wait[i] == 0]
        heapq.heapify(heap)
        shuffled = []
        while heap:
            _, i = heapq.heappop(heap)
            shuffled.append(i)
            for u in users[i]:
                n_wait[u] -= 1
                if n_wait[u] == 0:
                    heapq.heappush(heap, (prio[u], u))
        assert len(shuffled) == len(creation) and shuffled != creation
        outs, _, _ = drv.push_circuit_by_handles(pool, rec, order=shuffled)
        assert all(np.array_equal(x, y) for x, y in zip(outs, g_outs))
        # four pushers and a blocking 32-worker walk of the same circuit at the same time on the same pool: every one of them
        # gets the graph's words (the pushers' deferred batches mix operations of all four; the blocking callers' batches are
        # the ordinary lanes')
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=5) as ex:
            jobs = [ex.submit(drv.push_circuit_by_handles, pool, rec) for _ in range(4)]
            jobs.append(ex.submit(drv.run_circuit_by_handles, pool, rec, 32))
            for j in jobs:
                o = j.result()[0]
                assert len(o) == 33 and all(np.array_equal(x, y) for x, y in zip(o, g_outs))
        import gc
        gc.collect()
        vs = pool.value_stats()
        assert vs["live_values"] == 0 and vs["live_bytes"] == 0, vs
    finally:
        pool.close()
        g.close()


def _adder_inputs(ks, r, a, b):
    P = ks.params
    cts = []
    for i in range(32):
        for bit in ((a >> i) & 1, (b >> i) & 1):
            m = np.zeros(P.N, dtype=np.uint64)
            m[0] = O.encode(bit, 1)
            cts.append(O.encrypt_glwe(r, ks.glwe_sk, m, P.N, P.k, P.glwe_std))
    return np.stack(cts)


def test_reference_adder_op_by_op_from_64_threads_by_handles():
    """`mux_circuits::add::ripple_carry_adder(32, 32, false)` fed the way `add_circuit` feeds it (circuits/add.rs:10-32), every
    node ONE spf_pool_submit_op_v + spf_pool_wait from a pool of 64 native workers that walk the DAG as the reference's
    processor does (tools/pool_driver.cpp: spf_circuit_drive): decrypts to a + b, every output word-equal to spf_graph_run of
    the same circuit; a sample of outputs and one whole conversion against the oracle; afterwards no value is left alive."""
    import tools.driver as drv
    from spf_amd.gate_pool import circuit_jobs_as_one_graph
    from spf_amd.mux_circuits import ripple_carry_adder
    ks = keyset(0x5EED0001, 637)
    P = ks.params
    r = O.Rng(0xADD34)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    ak = O.gen_auto_key_fft(r, ks.glwe_sk, P)
    ssk = O.gen_ssk_fft(r, ks.glwe_sk, P)
    eng.load_automorphism_key(ak)
    eng.load_scheme_switch_key(ssk)
    adder = ripple_carry_adder(32, 32, False)
    a, b = 0xC0FFEE37, 0x7E3779B9
    cts = _adder_inputs(ks, r, a, b)
    rec, outs = circuit_jobs_as_one_graph(eng, adder, cts[None], record=True)
    assert len(rec.outputs) == 33
    # the same DAG as ONE gate graph
    g, g_outs = rec.lower(eng)
    g.run()
    pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=20)
    try:
        got, seconds, _ = drv.run_circuit_by_handles(pool, rec, threads=64)
        c = pool.counters()
        stats = pool.value_stats()
    finally:
        pool.close()
    g.close()
    n_tasks = sum(1 for o in rec.op if o >= 0)
    assert c["handle_ops"] == n_tasks == 64 * 3 + 1679 and c["ops"] == n_tasks, c
    assert c["handle_launches"] < n_tasks / 4, c            # the per-operation calls were coalesced
    assert stats["live_values"] == 0, stats                  # every intermediate was released by its last consumer
    total = 0
    for i in range(33):
        assert np.array_equal(got[i], g_outs[i]), f"output bit {i}: by handles differs from spf_graph_run"
        total |= O.decode(int(O.decrypt_glwe_raw(got[i], ks.glwe_sk, P.N, P.k)[0]), 1) << i
    assert total == a + b
    # against the oracle: the conversion of input bit 0 and the first sum bit's gates (sum_0 = a0 xor b0: a two-level tree)
    l1 = O.sample_extract(cts[0], 0, P.N, P.k)
    l0 = O.keyswitch_lwe(l1, ks.ksk, P.N, P.lwe_n, P.ks_radix_log, P.ks_count)
    sel_a0 = O.circuit_bootstrap(l0, ks.bsk_fft, ak, ssk, P)
    l0b = O.keyswitch_lwe(O.sample_extract(cts[1], 0, P.N, P.k), ks.ksk, P.N, P.lwe_n, P.ks_radix_log, P.ks_count)
    sel_b0 = O.circuit_bootstrap(l0b, ks.bsk_fft, ak, ssk, P)
    one = np.zeros(P.glwe_len, dtype=np.uint64)
    one[P.N * P.k] = 1 << 63
    zero = np.zeros(P.glwe_len, dtype=np.uint64)
    # ROBDD of a0 xor b0 over the order (a0, b0): root tests a0; low child = b0 ? 1 : 0, high child = b0 ? 0 : 1
    lo = O.cmux(zero, one, sel_b0, P.N, P.k, P.cbs_radix_log, P.cbs_count)
    hi = O.cmux(one, zero, sel_b0, P.N, P.k, P.cbs_radix_log, P.cbs_count)
    exp0 = O.cmux(lo, hi, sel_a0, P.N, P.k, P.cbs_radix_log, P.cbs_count)
    assert np.array_equal(got[0], exp0), "sum bit 0: by handles differs from the oracle"
    print(f"adder by handles: {n_tasks} operations in {c['handle_launches']} launches from 64 threads, {seconds * 1e3:.1f} ms")


def test_released_values_give_their_memory_back(rig):
    """hipMemGetInfo before / after: values created, used and released -> spf_pool_trim -> the free memory is back at the baseline;
    while they live, the arena caches and reuses blocks (no hipMalloc on the steady-state path)."""
    ks, eng, _ = rig
    P = ks.params
    hip = C.CDLL("libamdhip64.so")   # the runtime the library itself is linked against (already loaded)

    def free_bytes(at_least=0):
        # (the runtime hands freed memory back to the driver lazily: right after a thousand hipFree calls hipMemGetInfo may still
        # show them — 250 MiB of 262 in one run, gone half a second later, scratch probe in profiles/r06_values.md — so a
        # reading that has to reach a level is polled for up to five seconds)
        import time
        for _ in range(100):
            assert hip.hipDeviceSynchronize() == 0
            free, total = C.c_size_t(), C.c_size_t()
            assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
            if free.value >= at_least:
                break
            time.sleep(0.05)
        return free.value

    N_BALLAST = 4096
    pool = spf_amd.Pool(eng, max_batch=64, max_wait_us=200)
    big = np.random.default_rng(5).standard_normal((8, eng.params.cbs_ggsw_complex * 2)).view(np.complex128)
    try:
        def round_trip():
            vals = [pool.upload(ValueKind.LWE1, x) for x in random_lwe_batch(11, 16, P.N * P.k)]
            with ThreadPoolExecutor(max_workers=16) as ex:
                sels = list(ex.map(pool.keyswitch_circuit_bootstrap_v, vals))
            glwe = [pool.upload(ValueKind.GLWE1, x) for x in random_glwe(12, 16, P.glwe_len)]
            with ThreadPoolExecutor(max_workers=16) as ex:
                outs = list(ex.map(lambda i: pool.run_v(FheOp.CMux, [sels[i], glwe[i], glwe[(i + 1) % 16]]), range(16)))
            ballast = [pool.upload(ValueKind.GGSW1, big[i % 8]) for i in range(N_BALLAST)]   # 1 GiB of selectors alive at once
            peak = pool.value_stats()
            ck = sum(int(o.download()[0]) for o in outs)
            for v in vals + sels + glwe + outs + ballast:
                v.release()
            return ck, peak

        ck0, _ = round_trip()
        for _ in range(5):          # staging sets, streams and the runtime's per-queue scratch: allocated once
            round_trip()
        pool.trim()
        import time
        time.sleep(1.0)             # (see free_bytes: the frees of the trim settle)
        free0 = free_bytes()
        assert pool.value_stats() == {"live_values": 0, "live_bytes": 0, "cached_bytes": 0}
        for _ in range(2):
            ck, peak = round_trip()
            assert ck == ck0
            assert peak["live_values"] == 16 * 4 + N_BALLAST and peak["live_bytes"] >= N_BALLAST * 256 * 1024, peak
        mallocs = pool.counters()["value_mallocs"]
        assert round_trip()[0] == ck0
        # steady state: the uploads came from the cache; a batch of a size class not seen before may still allocate
        assert pool.counters()["value_mallocs"] <= mallocs + 8
        s = pool.value_stats()
        assert s["live_values"] == 0 and s["live_bytes"] == 0 and s["cached_bytes"] >= N_BALLAST * 256 * 1024, s
        held = free_bytes()
        pool.trim()
        free1 = free_bytes(at_least=held + s["cached_bytes"] - (32 << 20))
        assert pool.value_stats()["cached_bytes"] == 0
        # trim returns what the cache held (the driver accounts in 2 MiB granules: small blocks do not add up exactly) ...
        assert abs(free1 - held - s["cached_bytes"]) <= (32 << 20), (held, free1, s)
        # ... which is the baseline — up to what the HIP runtime itself keeps per hardware queue once a staging set's stream has run
        # a kernel with a private segment (tens of MiB, measured: profiles/r06_values.md); the values that went through are 1 GiB
        assert free0 - free1 <= (64 << 20), (free0, free1)
    finally:
        pool.close()


def test_values_on_a_group_pool_stay_on_their_member(rig):
    """Two members on the one GPU: a value lives on one member, an operation runs where its operands live, mixing members is an
    error, spf_value_copy_to_member moves a copy; results are word-equal to one context."""
    ks, eng, ssk = rig
    P = ks.params
    r = O.Rng(0x7A11)
    grp = spf_amd.Group(eng.params, devices=(0, 0))
    grp.load_bootstrap_key(ks.bsk_fft)
    grp.load_keyswitch_key(ks.ksk)
    grp.load_automorphism_key(O.gen_auto_key_fft(r, ks.glwe_sk, P))
    grp.load_scheme_switch_key(ssk)
    pool = spf_amd.Pool(grp, max_batch=64, max_wait_us=200)
    try:
        lwe1 = random_lwe_batch(21, 2, P.N * P.k)
        a = random_glwe(22, 2, P.glwe_len)
        b = random_glwe(23, 2, P.glwe_len)
        exp_sel = eng.circuit_bootstrap(eng.keyswitch_lwe_l1_lwe_l0(lwe1))
        exp = eng.cmux(exp_sel, a, b)
        outs = []
        for m in (0, 1):
            vl = pool.upload(ValueKind.LWE1, lwe1[m], member=m)
            va, vb = pool.upload(ValueKind.GLWE1, a[m], member=m), pool.upload(ValueKind.GLWE1, b[m], member=m)
            sel = pool.keyswitch_circuit_bootstrap_v(vl)
            assert sel.info()["member"] == m
            out = pool.run_v(FheOp.CMux, [sel, va, vb])
            assert out.info()["member"] == m
            assert np.array_equal(out.download(), exp[m])
            outs.append((sel, va, vb))
        sel0, va0, vb0 = outs[0]
        sel1, va1, vb1 = outs[1]
        with pytest.raises(spf_amd.SpfError):
            pool.submit_v(FheOp.CMux, [sel0, va1, vb1])    # operands on different members
        moved = pool.copy_to_member(sel0, 1)
        assert moved.info()["member"] == 1
        got = pool.run_v(FheOp.CMux, [moved, va1, vb1]).download()
        assert np.array_equal(got, eng.cmux(exp_sel[:1], a[1:], b[1:])[0])
        # pushed chains (pending operands) live on the member their operands live on, both members at the same time
        pushed = []
        for m, (va_m, vb_m) in enumerate(((va0, vb0), (va1, vb1))):
            vl = pool.upload(ValueKind.LWE1, lwe1[m], member=m)
            sel = pool.push_v(FheOp.CircuitBootstrap, [pool.push_v(FheOp.KeyswitchL1toL0, [vl])])
            pushed.append(pool.push_v(FheOp.CMux, [sel, pool.push_v(FheOp.Not, [va_m]), vb_m]))
        for m, v in enumerate(pushed):
            assert v.wait().info()["member"] == m
            assert np.array_equal(v.download(), eng.cmux(exp_sel[m:m + 1], eng.glwe_not(a[m:m + 1]), b[m:m + 1])[0])
        del pushed, sel, vl
    finally:
        import gc
        del outs
        gc.collect()
        pool.close()
        grp.close()


def test_batch_upload_and_download_and_a_failing_batch(rig):
    """`spf_value_upload_batch` / `spf_value_download_batch`: the bits of an integer in one block and one copy (consecutive values)
    or one gathered copy (scattered values), same words as one by one; and a batch that FAILS (no keyswitch key) reports the
    status to its waiter and leaves a result value that can never be used."""
    ks, eng, _ = rig
    P = ks.params
    pool = spf_amd.Pool(eng, max_batch=64, max_wait_us=200)
    try:
        x = random_glwe(77, 9, P.glwe_len)
        vals = pool.upload_batch(ValueKind.GLWE1, x)
        assert len(vals) == 9 and all(v.info()["valid"] for v in vals)
        p0 = vals[0].device_ptr()
        assert [v.device_ptr() - p0 for v in vals] == [i * P.glwe_len * 8 for i in range(9)]          # one block, consecutive
        assert np.array_equal(pool.download_batch(vals), x)                                            # one copy
        order = [8, 2, 5, 0, 7]
        assert np.array_equal(pool.download_batch([vals[i] for i in order]), x[order])                # gathered on the device
        assert np.array_equal(pool.download_batch([vals[3], vals[1]]), x[[3, 1]])                     # (two: one copy each)
        nots = [pool.run_v(FheOp.Not, [v]) for v in vals[:4]]                                          # results of four batches
        exp = np.stack([O.glwe_not(x[i], P.N, P.k) for i in range(4)])
        assert np.array_equal(pool.download_batch(nots), exp)
        with pytest.raises(spf_amd.SpfError):
            pool.download_batch([vals[0], pool.upload(ValueKind.LWE1, random_lwe_batch(1, 1, P.N * P.k)[0])])   # mixed kinds
    finally:
        import gc
        gc.collect()
        pool.close()
    bare = spf_amd.Engine(eng.params)          # a context without keys
    pool = spf_amd.Pool(bare, max_batch=8, max_wait_us=100)
    try:
        v = pool.upload(ValueKind.LWE1, random_lwe_batch(2, 1, P.N * P.k)[0])
        res, ticket = pool.submit_v(FheOp.KeyswitchL1toL0, [v])
        with pytest.raises(spf_amd.SpfError) as e:
            pool.wait(ticket)
        assert e.value.status == 3                                 # SPF_ERR_NO_KEY, the batch's status
        assert res.info()["valid"] is False
        with pytest.raises(spf_amd.SpfError):
            res.download()
        with pytest.raises(spf_amd.SpfError):
            pool.submit_v(FheOp.CircuitBootstrap, [res])           # a failed result is refused as an operand
        res.release()
        assert pool.run_v(FheOp.Not, [pool.upload(ValueKind.GLWE1, random_glwe(3, 1, P.glwe_len)[0])]).info()["valid"]   # the pool goes on
    finally:
        import gc
        gc.collect()
        pool.close()
        bare.close()
