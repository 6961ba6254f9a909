"""Device groups behind the C ABI (include/spf_hip.h, `spf_group_*`): one host process, a context per listed device,
keys replicated inside the library, a host batch cut into contiguous ranges of ceil(B / G).

The caller being replaced is ONE process — `Evaluation` holding an `Arc<ComputeKey>` (crypto/evaluation.rs:144-197),
called from the rayon workers of one `CircuitProcessor` (circuit_processor/mod.rs:201-209).  On a one-GPU box the group
is exercised with `[0]` and `[0, 0]` (two contexts on the one GPU: replication, split and reassembly all run); every
result must equal the single-context result word for word, and the oracle at n = 637.
"""
import os
import threading

import numpy as np
import pytest

import oracle as O
import spf_amd
from tests.util import keyset, random_glwe, random_lwe_batch, to_engine_params

pytestmark = pytest.mark.gpu

HOST_THREADS = max(1, min(16, os.cpu_count() or 1))
SMALL_N = 12


def _tail_keys(ks):
    r = O.Rng(0x7A11)
    return O.gen_auto_key_fft(r, ks.glwe_sk, ks.params), O.gen_ssk_fft(r, ks.glwe_sk, ks.params)


@pytest.fixture(scope="module")
def small():
    """n = 12: all four keys, one context and a group of two contexts on the one GPU"""
    ks = keyset(0x5EED0001, SMALL_N)
    ak, ssk = _tail_keys(ks)
    P = to_engine_params(ks.params)
    eng = spf_amd.Engine(P)
    grp = spf_amd.Group(P, devices=[0, 0])
    for e in (eng, grp):
        e.load_bootstrap_key(ks.bsk_fft)
        e.load_keyswitch_key(ks.ksk)
        e.load_automorphism_key(ak)
        e.load_scheme_switch_key(ssk)
    return ks, ak, ssk, eng, grp


def test_group_replicates_keys_in_library(small):
    ks, ak, ssk, eng, grp = small
    st = grp.replication_stats()
    P = grp.params
    assert len(grp) == 2 and grp.members_in_rotation() == 2
    # two contexts on ONE device: a one-rank RCCL communicator (ncclCommInitAll + in-place ncclBroadcast ran), then a
    # device-to-device copy for the second context
    assert st["transport"] == "rccl" and st["rccl_world_size"] == 1
    assert st["bytes_per_member"] == P.bsk_complex * 16 + P.ksk_words * 8 + ak.size * 16 + ssk.size * 16
    # the replica is the key: member 1 alone reproduces the single-context result
    lwe1 = random_lwe_batch(0x6A01, 5, ks.params.N)
    assert np.array_equal(grp.member(1).keyswitch_circuit_bootstrap(lwe1).view(np.float64),
                          eng.keyswitch_circuit_bootstrap(lwe1).view(np.float64))


@pytest.mark.parametrize("B", [1, 2, 7, 33])
def test_group_batches_equal_single_context(small, B):
    """every host-pointer entry point: split over [0, 0], reassembled, word-equal to one context (ragged B included:
    B = 1 leaves member 1 idle, B = 7 gives 4 + 3)"""
    ks, ak, ssk, eng, grp = small
    P = ks.params
    lwe1 = random_lwe_batch(0x6B00 + B, B, P.N)
    lwe0 = random_lwe_batch(0x6B40 + B, B, SMALL_N)
    glwe = random_glwe(0x6B80 + B, B, P.glwe_len)
    glwe2 = random_glwe(0x6BC0 + B, B, P.glwe_len)
    glev = random_glwe(0x6C00 + B, B * P.cbs_count, P.glwe_len).reshape(B, -1)
    glev2 = random_glwe(0x6C40 + B, B * P.cbs_count, P.glwe_len).reshape(B, -1)
    rng = np.random.default_rng(B)
    ggsw = (rng.standard_normal((B, grp.params.cbs_ggsw_complex * 2)) * 2.0 ** 60).view(np.complex128)
    lut = spf_amd.generate_lut([[0, 1]], 1, grp.params)
    luts = random_glwe(0x6C80 + B, B, P.glwe_len)

    def same(a, b):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))

    same(grp.keyswitch_lwe_l1_lwe_l0(lwe1), eng.keyswitch_lwe_l1_lwe_l0(lwe1))
    same(grp.circuit_bootstrap_pbs(lwe0), eng.circuit_bootstrap_pbs(lwe0))
    same(grp.generalized_pbs(lwe0, lut, 0, 1, 1 << 61), eng.generalized_pbs(lwe0, lut, 0, 1, 1 << 61))
    same(grp.generalized_pbs(lwe0, luts, 0, 0, 0), eng.generalized_pbs(lwe0, luts, 0, 0, 0))   # per-ciphertext LUTs: the stride follows the range
    same(grp.pbs_univariate(lwe0, lut), eng.pbs_univariate(lwe0, lut))
    same(grp.gate_bootstrap(lwe1), eng.gate_bootstrap(lwe1))
    same(grp.circuit_bootstrap(lwe0), eng.circuit_bootstrap(lwe0))
    same(grp.keyswitch_circuit_bootstrap(lwe1), eng.keyswitch_circuit_bootstrap(lwe1))
    same(grp.mod_switch_trace_and_rotate(glwe), eng.mod_switch_trace_and_rotate(glwe))
    same(grp.scheme_switch(glev), eng.scheme_switch(glev))
    same(grp.sample_extract_l1(glwe, 5), eng.sample_extract_l1(glwe, 5))
    same(grp.glwe_not(glwe), eng.glwe_not(glwe))
    same(grp.glwe_xor(glwe, glwe2), eng.glwe_xor(glwe, glwe2))
    same(grp.glwe_mul_xn(glwe, 1234), eng.glwe_mul_xn(glwe, 1234))
    same(grp.cmux(ggsw, glwe, glwe2), eng.cmux(ggsw, glwe, glwe2))
    same(grp.glev_cmux(ggsw, glev, glev2), eng.glev_cmux(ggsw, glev, glev2))
    same(grp.multiply_glwe_ggsw(glwe, ggsw), eng.multiply_glwe_ggsw(glwe, ggsw))


def test_group_constants_and_bincode_key(small):
    ks, ak, ssk, eng, grp = small
    for bit in (0, 1):
        assert np.array_equal(grp.l1ggsw_constant(bit).view(np.float64), eng.l1ggsw_constant(bit).view(np.float64))
    # the ComputeKey wire format through the group loader: parsed on member 0, replicated, same results
    blob = b"".join(np.uint64(a.size).tobytes() + np.ascontiguousarray(a).tobytes()
                    for a in (ks.bsk_fft.reshape(-1), ks.ksk.reshape(-1), ssk.reshape(-1), ak.reshape(-1)))
    g2 = spf_amd.Group(grp.params, devices=[0, 0, 0])
    try:
        g2.load_compute_key_bincode(blob)
        lwe1 = random_lwe_batch(0x6D01, 8, ks.params.N)
        assert np.array_equal(g2.keyswitch_circuit_bootstrap(lwe1).view(np.float64), eng.keyswitch_circuit_bootstrap(lwe1).view(np.float64))
        with pytest.raises(spf_amd.SpfError):
            g2.load_compute_key_bincode(blob[:-9])   # truncated: refused, nothing half-loaded is used
    finally:
        g2.close()


def test_group_failed_member_is_requeued(small):
    """a failed GPU's shard is re-queued by the host.  Member 1's next call fails (injected SPF_ERR_HIP): it
    leaves the rotation, its range runs on member 0, the caller sees the complete, correct result."""
    ks, ak, ssk, eng, grp = small
    lwe1 = random_lwe_batch(0x6E01, 9, ks.params.N)
    want = eng.gate_bootstrap(lwe1)
    grp.debug_fail_next(1, 1)
    assert np.array_equal(grp.gate_bootstrap(lwe1), want)
    assert grp.members_in_rotation() == 1
    assert np.array_equal(grp.gate_bootstrap(lwe1), want)        # one member left: still serves
    grp.set_member_enabled(1, True)                              # re-admitted by hand
    assert grp.members_in_rotation() == 2
    # a caller's error is not a device failure: returned, nobody leaves the rotation
    with pytest.raises(spf_amd.SpfError):
        grp.sample_extract_l1(random_glwe(1, 2, ks.params.glwe_len), ks.params.N)
    assert grp.members_in_rotation() == 2
    # drained by hand, and the last member failing is an error, not a hang
    grp.set_member_enabled(0, False)
    grp.debug_fail_next(1, 1)
    with pytest.raises(spf_amd.SpfError):
        grp.gate_bootstrap(lwe1)
    grp.set_member_enabled(0, True)
    grp.set_member_enabled(1, True)
    assert np.array_equal(grp.gate_bootstrap(lwe1), want)


def test_group_calls_from_many_threads(small):
    """the rayon shape: many host threads in the group at once, each with its own batch"""
    ks, ak, ssk, eng, grp = small
    lwes = [random_lwe_batch(0x6F00 + t, 3 + t % 4, ks.params.N) for t in range(12)]
    want = [eng.gate_bootstrap(x) for x in lwes]
    got, errs = [None] * len(lwes), []

    def run(t):
        try:
            got[t] = grp.gate_bootstrap(lwes[t])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(t,)) for t in range(len(lwes))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_group_pool_deals_callers_across_members(small):
    """spf_pool_create_group: single-ciphertext callers from many threads, dealt over the members; every caller's output
    is its own"""
    ks, ak, ssk, eng, grp = small
    P = grp.params
    T = 24
    lwe1 = random_lwe_batch(0x7001, T, ks.params.N)
    want = eng.keyswitch_circuit_bootstrap(lwe1)
    pool = spf_amd.Pool(grp, max_batch=64, max_wait_us=500)
    outs = [np.zeros(P.cbs_ggsw_complex, dtype=np.complex128) for _ in range(T)]
    errs = []

    def run(t):
        try:
            for _ in range(2):
                pool.keyswitch_circuit_bootstrap(outs[t], lwe1[t])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for t in range(T):
        assert np.array_equal(outs[t].view(np.float64), want[t].view(np.float64)), t
    ops, launches = pool.stats()
    assert ops == 2 * T and launches >= 2            # both members launched
    # a ticket of a member that does not exist is an error, not a crash
    with pytest.raises(spf_amd.SpfError):
        pool._wait((7 << 56) | 1)
    pool.close()


def test_group_of_one_and_forced_rccl():
    """a group of [0] is one context behind the group entry points; with SPF_GROUP_TRANSPORT=rccl even the one-member
    group opens its communicator and broadcasts in place (the RCCL path at world size 1)"""
    ks = keyset(0x5EED0001, SMALL_N)
    P = to_engine_params(ks.params)
    lwe1 = random_lwe_batch(0x7101, 6, ks.params.N)
    eng = spf_amd.Engine(P)
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    want = eng.gate_bootstrap(lwe1)
    old = os.environ.get("SPF_GROUP_TRANSPORT")
    try:
        for transport, world in ((None, 0), ("rccl", 1), ("peer", 0)):
            if transport is None:
                os.environ.pop("SPF_GROUP_TRANSPORT", None)
            else:
                os.environ["SPF_GROUP_TRANSPORT"] = transport
            g = spf_amd.Group(P, devices=[0])
            g.load_bootstrap_key(ks.bsk_fft)
            g.load_keyswitch_key(ks.ksk)
            st = g.replication_stats()
            assert st["rccl_world_size"] == world and st["transport"] == (transport or "none"), st
            assert np.array_equal(g.gate_bootstrap(lwe1), want)
            g.close()
        os.environ["SPF_GROUP_TRANSPORT"] = "carrier-pigeon"
        with pytest.raises(spf_amd.SpfError):
            spf_amd.Group(P, devices=[0])
    finally:
        if old is None:
            os.environ.pop("SPF_GROUP_TRANSPORT", None)
        else:
            os.environ["SPF_GROUP_TRANSPORT"] = old
    # peer transport with two contexts on one device: the second one takes the device-to-device copy
    os.environ["SPF_GROUP_TRANSPORT"] = "peer"
    try:
        g = spf_amd.Group(P, devices=[0, 0])
        g.load_bootstrap_key(ks.bsk_fft)
        g.load_keyswitch_key(ks.ksk)
        assert np.array_equal(g.gate_bootstrap(lwe1), want)
        g.close()
    finally:
        os.environ.pop("SPF_GROUP_TRANSPORT", None)
        if old is not None:
            os.environ["SPF_GROUP_TRANSPORT"] = old
    with pytest.raises(spf_amd.SpfError):
        spf_amd.Group(P, devices=[0, 99])          # no such device: refused, nothing leaks
    with pytest.raises(spf_amd.SpfError):
        spf_amd.Group(P, devices=[])


def test_group_default128_against_the_oracle():
    """n = 637, ragged B over [0, 0]: keys loaded through the group, every output against the oracle and against one
    context (the split must not change which kernel shape sees which ciphertext's bits)"""
    ks = keyset(0x5EED0001, 637)
    P = to_engine_params(ks.params)
    grp = spf_amd.Group(P, devices=[0, 0])
    grp.load_bootstrap_key(ks.bsk_fft)
    grp.load_keyswitch_key(ks.ksk)
    B = 1031          # 516 + 515: both halves on the two-per-workgroup shape, the second one ragged
    lwe = random_lwe_batch(0x7201, B, 637)
    got = grp.circuit_bootstrap_pbs(lwe)
    _, exp = O.bench_cbs_pbs(lwe, ks.bsk_fft, ks.params, HOST_THREADS, native=False)
    bad = np.nonzero((got != exp).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} ciphertexts differ, first {bad[:8]}"
    lwe1 = random_lwe_batch(0x7202, 9, ks.params.N)
    gate = grp.gate_bootstrap(lwe1)
    for i in range(9):
        l0 = O.keyswitch_lwe(lwe1[i], ks.ksk, ks.params.N, 637, ks.params.ks_radix_log, ks.params.ks_count)
        assert np.array_equal(gate[i], O.cbs_pbs(l0, ks.bsk_fft, ks.params)), i
    grp.close()


def test_gate_graph_jobs_are_dealt_over_the_members(small):
    """VERDICT r05 ⊕x6: config 5's gate pool from ONE process through the C ABI.  Four 8 x 8 multiplications through the
    reference's multiplier block (mux_circuits `unsigned_multiplier(8, 8)`, 3 228 CMUX each) as four jobs of a group [0, 0]:
    `spf_group_run_graphs` deals them 2 + 2 (equal cost, longest processing time first), lowers each member's two jobs into one
    graph and runs the members side by side; every product decrypts, every output word equals the same job run as an ordinary
    graph on one context; a fifth, smaller job goes to the member with less load; a failing member's jobs are dealt again; a
    destroyed job leaves no merged graph behind."""
    from spf_amd.gate_pool import circuit_jobs_as_one_graph
    from spf_amd.mux_circuits import parse_mux_circuit, ripple_carry_adder
    ks, ak, ssk, eng, grp = small
    P = ks.params
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    circuit = parse_mux_circuit(open(os.path.join(root, "spf_amd", "data", "mux_multiplier_n8_m8.bincode"), "rb").read())
    r = O.Rng(0x6A0B)
    pairs = [(0xB7, 0x5D), (255, 255), (3, 200), (0x80, 0x81)]

    def encrypt_bits(bits):
        cts = []
        for bit in bits:
            m = np.zeros(P.N, dtype=np.uint64)
            m[0] = O.encode(bit, 1)
            cts.append(O.encrypt_glwe(r, ks.glwe_sk, m, P.N, P.k, P.glwe_std))
        return np.stack(cts)

    inputs = [encrypt_bits([(a >> i) & 1 for i in range(8)] + [(b >> i) & 1 for i in range(8)]) for a, b in pairs]
    jobs, outs = [], []
    for x in inputs:
        g, o = circuit_jobs_as_one_graph(grp, circuit, x[None])
        jobs.append(g)
        outs.append(o[0])
    assert all(g.member() == -1 for g in jobs)
    grp.run_graphs(jobs)
    where = [g.member() for g in jobs]
    assert sorted(where) == [0, 0, 1, 1], where           # dealt 2 + 2
    st = jobs[0].stats()
    assert st["nodes"] == 16 * 4 + 3228 + 2 and st["levels"] == 3 + 126
    for (a, b), x, o in zip(pairs, inputs, outs):
        got = 0
        for i in range(16):
            got |= O.decode(int(O.decrypt_glwe_raw(o[i], ks.glwe_sk, P.N, P.k)[0]), 1) << i
        # (n = 12 is not a secure key switch target, but the arithmetic is the same: the product decrypts)
        assert got == a * b, (a, b, got)
        g1, o1 = circuit_jobs_as_one_graph(eng, circuit, x[None])   # the same job on ONE context
        g1.run()
        for i in range(16):
            assert np.array_equal(o[i], o1[0][i]), (a, b, i)
        g1.close()
    first = [np.copy(v) for v in outs[0]]
    grp.run_graphs(jobs)                                            # again: the merged graphs are reused, same words
    assert all(np.array_equal(u, v) for u, v in zip(first, outs[0]))
    # a fifth, cheaper job (an 8-bit adder): the costs decide — heaviest first, the adder lands beside two multipliers
    adder = ripple_carry_adder(8, 8, False)
    xa = encrypt_bits([1, 0] * 8)
    ga, oa = circuit_jobs_as_one_graph(grp, adder, xa[None])
    grp.run_graphs(jobs + [ga])
    assert sorted(g.member() for g in jobs) == [0, 0, 1, 1] and ga.member() in (0, 1)
    total = 0
    for i, o in enumerate(oa[0]):
        total |= O.decode(int(O.decrypt_glwe_raw(o, ks.glwe_sk, P.N, P.k)[0]), 1) << i
    assert total == 0xFF                                            # a = 0xFF (the even inputs), b = 0
    # a member that fails is taken out of rotation and its jobs run on the other one
    grp.debug_fail_next(1, 1)
    grp.run_graphs(jobs)
    assert [g.member() for g in jobs] == [0, 0, 0, 0] and grp.members_in_rotation() == 1
    assert all(np.array_equal(u, v) for u, v in zip(first, outs[0]))
    grp.set_member_enabled(1, True)
    # a job that is destroyed leaves no merged graph reading its buffers; the others still run
    jobs[3].close()
    ga.close()
    grp.run_graphs(jobs[:3])
    assert all(np.array_equal(u, v) for u, v in zip(first, outs[0]))
    with pytest.raises(spf_amd.SpfError):                           # a graph of one context is not a job of the group
        g1, _ = circuit_jobs_as_one_graph(eng, adder, xa[None])
        try:
            grp.run_graphs([g1])
        finally:
            g1.close()
    for g in jobs[:3]:
        g.close()
