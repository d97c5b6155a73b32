"""SURVEY 8(f)3, the last fold sites the review listed: SpartanOuterProver's standard rounds (src/zkvm/spartan/outer.zig:364-407),
Phase1Prover / initShiftQBuffers (src/zkvm/spartan/prefix_suffix.zig:35-232) and the Lasso PrefixPolynomial (src/zkvm/lasso/
prefix_suffix.zig:133-231). The restatements are pinned on the reference's OWN test vectors (prefix_suffix.zig "Phase1Prover basic":
P = [1,2,3,4], Q = [5,6,7,8] -> (26, 44), bind 2 -> P = [3, 5]; lasso "prefix polynomial bind": [1,2,3,4] bound at 2 -> [5, 6]); the device
mirrors against the restatements on random tables."""
import numpy as np
import pytest

from tests import util as U


def _ob():
    from oracle import binding as ob
    return ob


def test_restatements_on_the_references_own_vectors():
    ob = _ob()
    f = lambda vals: np.stack([ob.fr_from_int(v) for v in vals])
    p = ob.Phase1Prover()
    p.addPair(f([1, 2, 3, 4]), f([5, 6, 7, 8]))
    assert p.computeRoundEvals() == [26, 44] and not p.shouldTransition()
    p.bind(2)
    assert p.current_size == 2 and p.pairs[0][0] == [3, 5] and p.shouldTransition()
    lp = ob.LassoPrefixPolynomial(f([1, 2, 3, 4])).bind(2)
    assert lp.num_vars == 1 and lp.evaluations == [5, 6]
    # evaluate: the index's low bit belongs to point[0] (prefix_suffix.zig:198-216)
    q = ob.LassoPrefixPolynomial(f([1, 2, 3, 4]))
    P = ob._R_P
    assert q.evaluate([0, 0]) == 1 and q.evaluate([1, 0]) == 2 and q.evaluate([0, 1]) == 3 and q.evaluate([1, 1]) == 4
    assert q.evaluate([5, 7]) == (1 * (1 - 5) * (1 - 7) + 2 * 5 * (1 - 7) + 3 * (1 - 5) * 7 + 4 * 35) % P
    # SpartanOuterProver rounds: p(2) = 2 p(1) - p(0); the fold (1 - r) even + r odd
    o = ob.SpartanOuterRounds(f([1, 2, 3, 4]))
    assert o.computeStandardRoundPoly() == [4, 6, 8]
    o.bindChallenge(3)
    assert o.vals == [(1 - 3) * 1 + 3 * 2, (1 - 3) * 3 + 3 * 4]
    o.bindChallenge(2)
    assert o.computeStandardRoundPoly() == [o.vals[0], 0, 0]
    # initShiftQBuffers against the whole-prover restatement's own Q tables (two routes to the same sums)
    n = 4
    T = 1 << n
    w = [[int(v) for v in row] for row in np.random.default_rng(5).integers(0, 1 << 40, size=(T, 43))]
    ro, rp = [3, 5, 7, 11], [13, 17, 19, 23]
    g = [pow(29, i, P) for i in range(5)]
    sh = ob.Stage3ShiftProver(w, ro, rp, g)
    col = lambda name: f([row[ob.R1CS_INPUT_NAMES.index(name)] for row in w])
    q = ob.init_shift_q_buffers(col("UnexpandedPC"), col("PC"), col("FlagVirtualInstruction"), col("FlagIsFirstInSequence"), col("FlagIsNoop"),
                                f(ob._s3_eq(ro[:2])), f(ob._s3_eqp1(ro[:2])), f(ob._s3_eq(rp[:2])), f(ob._s3_eqp1(rp[:2])), f(g), 4)
    assert [[ob.fr_to_int(x) for x in t] for t in q] == sh.Q


@pytest.mark.gpu
@pytest.mark.parametrize("v", [1, 2, 5, 10, 14])
def test_device_mirrors_of_the_remaining_fold_sites(v):
    ob = _ob()
    from zolt_amd import api, lib
    lib.init()
    n = 1 << v
    rnd = lambda seed, k: ob.f_to_mont(ob.FR, U.random_raw256(seed, k))
    ch = rnd(8100 + v, v + 1)
    # SpartanOuterProver
    tab = rnd(8000 + v, n)
    a, b = api.SpartanOuterProver(tab), ob.SpartanOuterRounds(tab)
    for k in range(v + 1):
        assert [ob.fr_to_int(x) for x in a.computeStandardRoundPoly()] == b.computeStandardRoundPoly(), k
        a.bindChallenge(ch[k])
        b.bindChallenge(ob.fr_to_int(ch[k]))
    a.deinit()
    # Phase1Prover with 1, 2, 3 and 5 pairs
    for npairs in (1, 2, 3, 5):
        a, b = api.Phase1Prover(), ob.Phase1Prover()
        for j in range(npairs):
            P, Q = rnd(8200 + 10 * v + j, n), rnd(8300 + 10 * v + j, n)
            a.addPair(P, Q)
            b.addPair(P, Q)
        for k in range(v):
            assert [ob.fr_to_int(x) for x in a.computeRoundEvals()] == b.computeRoundEvals(), (npairs, k)
            assert a.shouldTransition() == b.shouldTransition()
            a.bind(ch[k])
            b.bind(ob.fr_to_int(ch[k]))
        assert [[ob.fr_to_int(x[0]) for x in pq] for pq in a.pairs()] == [[pq[0][0], pq[1][0]] for pq in b.pairs]
        a.deinit()
    # Lasso PrefixPolynomial
    a, b = api.LassoPrefixPolynomial(tab), ob.LassoPrefixPolynomial(tab)
    if v <= 10:
        pt = rnd(8400 + v, v)
        assert ob.fr_to_int(a.evaluate(pt)) == b.evaluate([ob.fr_to_int(x) for x in pt])
    for k in range(v):
        a, b = a.bind(ch[k]), b.bind(ob.fr_to_int(ch[k]))
        assert [ob.fr_to_int(x) for x in a.evaluations] == b.evaluations, k
    # initShiftQBuffers
    if 2 <= v <= 12:
        ps = 1 << (v - v // 2)
        ss = n // ps
        cols = [rnd(8500 + 10 * v + j, n) for j in range(5)]
        suf = [rnd(8600 + 10 * v + j, ss) for j in range(4)]
        g = rnd(8700 + v, 5)
        got = api.initShiftQBuffers(*cols, *suf, g, ps)
        want = ob.init_shift_q_buffers(*cols, *suf, g, ps)
        for x, y in zip(got, want):
            assert np.array_equal(x, y)


@pytest.mark.gpu
def test_device_mirrors_on_the_references_own_vectors():
    ob = _ob()
    from zolt_amd import api, lib
    lib.init()
    f = lambda vals: np.stack([ob.fr_from_int(v) for v in vals])
    p = api.Phase1Prover()
    p.addPair(f([1, 2, 3, 4]), f([5, 6, 7, 8]))
    assert [ob.fr_to_int(x) for x in p.computeRoundEvals()] == [26, 44]
    p.bind(ob.fr_from_int(2))
    assert p.current_size == 2 and [ob.fr_to_int(x) for x in p.pairs()[0][0]] == [3, 5]
    p.deinit()
    lp = api.LassoPrefixPolynomial(f([1, 2, 3, 4])).bind(ob.fr_from_int(2))
    assert lp.num_vars == 1 and [ob.fr_to_int(x) for x in lp.evaluations] == [5, 6]


def _trace(seed, trace_len, k, start):
    rng = np.random.default_rng(seed)
    acc, init = [], {start + 8 * int(a): int(v) for a, v in zip(rng.integers(0, k, size=3), rng.integers(0, 1 << 40, size=3))}
    for ts in range(trace_len):
        if rng.random() < 0.6:
            addr = start + 8 * int(rng.integers(0, k + 2))  # a few addresses past the table: skipped by the reference
            acc.append((ts, addr, bool(rng.random() < 0.5), int(rng.integers(0, 1 << 62))))
    acc.append((trace_len + 3, start, True, 5))  # a timestamp past the trace: skipped
    acc.append((0, start - 8, True, 7))  # an address below the RAM region: skipped
    return acc, init


def test_lt_polynomial_is_the_less_than_indicator_on_the_cube():
    """the reference's own LtPolynomial test (val_evaluation.zig "lt polynomial basic"), extended: at a boolean point r the table is
    [j < r] for every index j (bit i of the integers <-> r[i])"""
    ob = _ob()
    for v in range(1, 5):
        for r in range(1 << v):
            bits = [(r >> i) & 1 for i in range(v)]
            assert ob.lt_table_int(bits) == [1 if j < r else 0 for j in range(1 << v)], (v, r)
    assert ob.lt_table_int([1, 0])[:3] == [1, 0, 0]  # the reference's vector: r_cycle = [1, 0] is cycle 1


@pytest.mark.gpu
@pytest.mark.parametrize("trace_len,log_k,log_t", [(1, 3, 1), (37, 4, 6), (256, 6, 8), (1000, 8, 10), (5000, 10, 13), (300, 5, 12), (300, 5, 7)])
def test_standard_stage4_val_evaluation(trace_len, log_k, log_t):
    """MultiStageProver.proveStage4 (prover.zig:713-828) through api.proveStage4 (tables: eq gather, zg_fr_lt_table; rounds: one product
    session) against the restatement, both on Keccak transcripts seeded alike: challenges, initial claim, every round polynomial, the final
    openings and claim; the transcripts end in the same state"""
    ob = _ob()
    from zolt_amd import api, lib
    lib.init()
    start = 0x80000000
    acc, init = _trace(9000 + trace_len, trace_len, 1 << log_k, start)
    ta, tb = api.Transcript(b"Jolt"), ob.Transcript(b"Jolt")
    ta.appendBytes(b"stage4" * 5)
    tb.append_bytes(b"stage4" * 5)
    got = api.proveStage4(acc, init, trace_len, log_k, log_t, start, ta)
    want = ob.stage4_prove(acc, init, trace_len, log_k, log_t, start, tb)
    assert np.array_equal(np.array(got["r_address"]), np.array(want["r_address"])) and np.array_equal(np.array(got["r_cycle"]), np.array(want["r_cycle"]))
    assert np.array_equal(got["initial_claim"], want["initial_claim"])
    assert len(got["round_polys"]) == len(want["round_polys"])
    for k, (a, b) in enumerate(zip(got["round_polys"], want["round_polys"])):
        assert np.array_equal(a, b), k
    assert np.array_equal(np.array(got["challenges"]).reshape(-1, 4), np.array(want["challenges"]).reshape(-1, 4))
    assert np.array_equal(got["final_claim"], want["final_claim"])
    assert bytes(ta.state) == tb.state_bytes()[0]
    # the tables themselves, and the device's lt table against the restatement's formula
    ra, rc = [ob.fr_to_int(x) for x in want["r_address"]], [ob.fr_to_int(x) for x in want["r_cycle"]]
    inc, wa, lt = api.valEvaluationTables(acc, init, trace_len, 1 << log_k, np.array(want["r_address"]), np.array(want["r_cycle"]), start)
    winc, wwa, wlt = ob.val_evaluation_tables(acc, init, trace_len, 1 << log_k, ra, rc, start)
    for x, y in ((inc, winc), (wa, wwa), (lt, wlt)):
        assert [ob.fr_to_int(v) for v in x] == y
    # inc and wa scattered on the device from the list of writes (zg_fr_write_tables_dev, what the compiled host's proveStage4 calls): the
    # list keeps the last write of a cycle; a second, unaligned write per few cycles is added to exercise exactly that
    acc2 = []
    for e in acc:
        acc2.append(e)
        if e[2] and e[0] % 3 == 0 and e[0] < trace_len:
            acc2.append((e[0], start + 8 * (e[0] % (1 << log_k)) + 1, True, e[3] ^ 0x5A5A))
    winc2, wwa2, _ = ob.val_evaluation_tables(acc2, init, trace_len, 1 << log_k, ra, rc, start)
    n, k = len(winc2), 1 << log_k
    last = {a: v for a, v in (init or {}).items() if a >= start and (a - start) // 8 < k}
    slot, rows = {}, []
    for ts, addr, is_write, value in acc2:
        if not is_write or addr < start or (addr - start) // 8 >= k or ts >= trace_len:
            continue
        row = [ts, (addr - start) // 8, last.get(addr, 0), value]
        if ts in slot:
            rows[slot[ts]] = row
        else:
            slot[ts] = len(rows)
            rows.append(row)
        last[addr] = value
    col = lambda i, dt: np.array([r[i] for r in rows], dtype=dt)
    d = lib.DeviceBuffer(2 * n * 32)
    lib.fr_write_tables_dev(n, col(0, np.uint32), col(1, np.uint32), col(2, np.uint64), col(3, np.uint64), np.array(want["r_address"])[::-1].copy(),
                            d.ptr, d.ptr + n * 32)
    both = d.to_host().view(np.uint64).reshape(2, n, 4)
    assert [ob.fr_to_int(v) for v in both[0]] == winc2 and [ob.fr_to_int(v) for v in both[1]] == wwa2
    if len(rows) >= 1:  # a list that names a cycle twice, or a cycle beyond the tables, is refused (the scatter would race / overrun)
        for bad in (np.array([rows[0][0], rows[0][0]], dtype=np.uint32), np.array([n], dtype=np.uint32)):
            z32, z64 = np.zeros(len(bad), dtype=np.uint32), np.zeros(len(bad), dtype=np.uint64)
            with pytest.raises(lib.ZgError):
                lib.fr_write_tables_dev(n, bad, z32, z64, z64, np.array(want["r_address"])[::-1].copy(), d.ptr, d.ptr + n * 32)
    d.free()


def test_expanding_table_restatement_on_the_references_vectors():
    """src/zkvm/lasso/expanding_table.zig tests: bind 3 -> [1 - 3, 3], sum 1; binds 2, 3, 5 -> 8 entries, sum 1, entry 0 = -8, entry 7 = 30;
    condense after binds 2, 3 under weights [1, 2, 3, 4] to one bit. (The reference's own expectation for entry 1 of the second test,
    r0 (1 - r1)(1 - r2) = 16, disagrees with its bind, which puts the LAST challenge on the low bit: entry 1 = (1 - r0)(1 - r1) r2 = 10 —
    the restatement follows the code.)"""
    ob = _ob()
    P = ob._R_P
    t = ob.ExpandingTable(3)
    assert t.values == [1]
    t.bind(3)
    assert t.values == [(1 - 3) % P, 3] and t.sum() == 1
    t = ob.ExpandingTable(4)
    for r in (2, 3, 5):
        t.bind(r)
    assert len(t.values) == 8 and t.sum() == 1 and t.values[0] == (-8) % P and t.values[7] == 30 and t.values[1] == 10
    t = ob.ExpandingTable(4)
    t.bind(2)
    t.bind(3)
    v = t.values
    assert t.condense([1, 2, 3, 4], 1) == [(v[0] * 1 + v[1] * 2) % P, (v[2] * 3 + v[3] * 4) % P]


@pytest.mark.gpu
@pytest.mark.parametrize("rounds", [1, 3, 8, 13])
def test_expanding_table_on_the_device(rounds):
    ob = _ob()
    from zolt_amd import api, lib
    lib.init()
    ch = ob.f_to_mont(ob.FR, U.random_raw256(8800 + rounds, rounds + 1))
    a, b = api.ExpandingTable(rounds, ch[rounds]), ob.ExpandingTable(rounds, ob.fr_to_int(ch[rounds]))
    for k in range(rounds):
        a.bind(ch[k])
        b.bind(ob.fr_to_int(ch[k]))
        assert [ob.fr_to_int(x) for x in a.getAll()] == b.values, k
    assert ob.fr_to_int(a.sum()) == b.sum()
    w = ob.f_to_mont(ob.FR, U.random_raw256(8900 + rounds, 1 << rounds))
    for out_bits in {0, rounds // 2, rounds}:
        assert [ob.fr_to_int(x) for x in a.condense(w, out_bits)] == b.condense([ob.fr_to_int(x) for x in w], out_bits), out_bits


def test_c_restatements_of_the_lt_table_and_the_column_sums():
    """oracle/zolt_oracle.c's zo_lt_table / zo_weighted_colsum (the CPU side of tools/crossover.py) against the Python restatements"""
    ob = _ob()
    r = [ob.fr_to_int(x) for x in ob.f_to_mont(ob.FR, U.random_raw256(9901, 7))]
    assert [ob.fr_to_int(x) for x in ob.lt_table(np.stack([ob.fr_from_int(v) for v in r]))] == ob.lt_table_int(r)
    tab = ob.f_to_mont(ob.FR, U.random_raw256(9902, 6 * 10))
    w = ob.f_to_mont(ob.FR, U.random_raw256(9903, 3 * 6)).reshape(3, 6, 4)
    got = ob.weighted_colsum(tab, 6, 10, w)
    t3 = tab.reshape(6, 10, 4)
    for k in range(3):
        for c in range(10):
            assert np.array_equal(got[k, c], ob._fsum(ob._fmul(np.ascontiguousarray(t3[:, c]), np.ascontiguousarray(w[k]))))


@pytest.mark.gpu
@pytest.mark.parametrize("v", [0, 1, 5, 11, 12, 13, 16, 17])  # 12 and above: the factored form (hi / lo factor tables, one product per entry)
def test_lt_table_and_column_sums_against_the_c_restatement(v):
    ob = _ob()
    from zolt_amd import lib
    lib.init()
    r = ob.f_to_mont(ob.FR, U.random_raw256(9910 + v, max(v, 1)))[:v]
    assert np.array_equal(lib.fr_lt_table(r), ob.lt_table(r))
    n = 1 << v
    rows = 1 << (v // 2)
    tab = ob.f_to_mont(ob.FR, U.random_raw256(9920 + v, n))
    w = ob.f_to_mont(ob.FR, U.random_raw256(9930 + v, 4 * rows)).reshape(4, rows, 4)
    assert np.array_equal(lib.fr_weighted_colsum(tab, rows, n // rows, w), ob.weighted_colsum(tab, rows, n // rows, w))
