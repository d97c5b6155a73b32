"""Runs the C++ host-mirror test binary (the reference's inline tests restated in C++ over the C ABI)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_host_mirror():
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    # always through make: a binary older than zolt_host.hpp or the header must not be the one that is tested
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "0 failures" in res.stdout


def _hexfr(x):
    return " ".join("%x" % int(v) for v in x)


def _write_rwc_case(path, log_k, log_t, p1, start, gamma, claim, r_cycle, initial_ram, accesses, challenges):
    with open(path, "w") as f:
        f.write(f"{log_k} {log_t} {p1} {start}\n{_hexfr(gamma)}\n{_hexfr(claim)}\n")
        for r in r_cycle:
            f.write(_hexfr(r) + "\n")
        f.write(f"{len(initial_ram)}\n" + "".join(f"{a} {v}\n" for a, v in initial_ram.items()))
        f.write(f"{len(accesses)}\n" + "".join(f"{ts} {a} {int(w)} {v}\n" for ts, a, w, v in accesses))
        for c in challenges:
            f.write(_hexfr(c) + "\n")


def _parse_rwc_output(text):
    import numpy as np
    rounds, claims, opening = [], [], None
    for line in text.splitlines():
        w = line.split()
        if not w:
            continue
        if w[0] == "E":
            rounds.append(np.array([int(x, 16) for x in w[1:17]], dtype=np.uint64).reshape(4, 4))
        elif w[0] == "C":
            claims.append((np.array([int(x, 16) for x in w[1:5]], dtype=np.uint64), int(w[5])))
        elif w[0] == "O":
            opening = np.array([int(x, 16) for x in w[1:13]], dtype=np.uint64).reshape(3, 4)
    return rounds, claims, opening


@pytest.mark.gpu
def test_cpp_ram_read_write_checking_mirror(tmp_path, golden_dir):
    """zolt::RamReadWriteCheckingProver (zolt_amd/host/zolt_host.hpp) — compiled host code over the C ABI — against the oracle's
    restatement of src/zkvm/ram/read_write_checking.zig on the reference's captured run and on random traces: every round polynomial,
    claim, entry count and the three opening claims."""
    import json
    import numpy as np
    from oracle import binding as ob
    from tests import util as U
    from tests.test_gpu_rwc import _trace
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])
    rwc = json.load(open(os.path.join(golden_dir, "rwc_captured_run.json")))
    stage2 = json.load(open(os.path.join(golden_dir, "stage2_batched_rounds.json")))
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    acc, gamma, r_cycle, initial_ram, challenges = U.rwc_inputs_of_the_captured_run(rwc, stage2, elf, ob.fr_from_int)
    cases = [(rwc["log_k"], rwc["log_t"], rwc["phase1_num_rounds"], rwc["start_address"], gamma, 0, r_cycle, initial_ram, acc, challenges)]
    for log_k, log_t, p1, n_acc in ((4, 8, 4, 200), (10, 13, 6, 3000), (3, 6, 0, 64)):
        start = 0x80000000
        a2, i2 = _trace(7000 + log_t, log_k, log_t, n_acc, start)
        g2 = ob.f_to_mont(ob.FR, U.random_raw256(70, 1))[0]
        rc2 = ob.f_to_mont(ob.FR, U.random_raw256(71 + log_t, log_t))
        ch2 = ob.f_to_mont(ob.FR, U.random_raw256(72 + log_t, log_k + log_t))
        cases.append((log_k, log_t, p1, start, g2, None, rc2, i2, a2, ch2))
    for k, (log_k, log_t, p1, start, g, claim, rc, init, accs, chal) in enumerate(cases):
        o = ob.RamReadWriteCheckingProver(accs, g, rc, log_k, log_t, p1, start, np.zeros(4, dtype=np.uint64), init)
        if claim is None:  # the true sum, so that every round is a sumcheck round
            P, gi = ob._R_P, ob.fr_to_int(g)
            claim = sum(ob.fr_to_int(o.eq_evals[e[0]]) * e[2] * (e[3] + gi * (e[3] + ob.fr_to_int(o.inc[e[0]]))) for e in o.entries) % P
        o.current_claim = claim
        path = str(tmp_path / f"rwc_{k}.txt")
        _write_rwc_case(path, log_k, log_t, p1, start, g, ob.fr_from_int(claim), rc, init, accs, chal)
        res = subprocess.run([exe, "rwc", path], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        rounds, claims, opening = _parse_rwc_output(res.stdout)
        assert len(rounds) == len(claims) == log_k + log_t
        for rd in range(log_k + log_t):
            we = o.computeRoundPolynomialCubic()
            assert np.array_equal(rounds[rd], we), (k, rd)
            o.updateClaim(we, chal[rd])
            o.bindChallenge(chal[rd])
            assert ob.fr_to_int(claims[rd][0]) == o.current_claim and claims[rd][1] == len(o.entries), (k, rd)
        wo = o.getOpeningClaims(chal)
        assert all(np.array_equal(opening[i], wo[i]) for i in range(3)), k
        if k == 0:  # the captured run: the reference's own final claim and opening claims, full width
            assert o.current_claim == int.from_bytes(bytes.fromhex(stage2["instance_final_claims"][2]), "little")
            assert ob.fr_to_int(opening[0]) == int(rwc["opening"]["ra_claim_be"], 16) and ob.fr_to_int(opening[1]) == int(rwc["opening"]["val_claim_be"], 16)


@pytest.mark.gpu
def test_cpp_stage4_registers_mirror(tmp_path, golden_dir):
    """zolt::Stage4GruenProver (compiled host code over zg_rrw_*) against the restatement of src/zkvm/spartan/stage4_gruen_prover.zig:
    the reference's captured run (its printed round-0 evaluations and final values, full width) and seeded traces, every round's four
    evaluations and the final claims bit for bit."""
    import json
    import numpy as np
    from oracle import binding as ob
    from tests import util as U
    from tests.test_gpu_stage4 import seeded_steps
    from tests.test_transcript_host import stage4_inputs_of_the_captured_run
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])
    fx, gr, steps, gamma, r_cycle = stage4_inputs_of_the_captured_run(golden_dir, ob.fr_from_int)
    le = lambda h: int.from_bytes(bytes.fromhex(h), "little")
    chal = np.stack([ob.fr_from_int(le(h)) for h in fx["challenges_le"][:15]])
    cases = [(8, fx["phase1_num_rounds"], gamma, ob.fr_from_int(le(fx["round0"]["claim_le"])), r_cycle, steps, chal)]
    for log_t, n_steps, p1 in ((3, 7, 2), (9, 400, 9), (11, 2048, 5)):
        r = ob.f_to_mont(ob.FR, U.random_raw256(900 + log_t, 2 * log_t + 8))
        cases.append((log_t, p1, r[0], None, r[1:1 + log_t], seeded_steps(40 + log_t, n_steps), r[1 + log_t:]))
    for k, (log_t, p1, g, claim, rc, st, ch) in enumerate(cases):
        o = ob.Stage4GruenProver(st, g, rc, p1, 7)
        if claim is None:
            claim = o.computeInputClaim()
        path = str(tmp_path / f"stage4_{k}.txt")
        with open(path, "w") as f:
            f.write(f"{log_t} {p1}\n{_hexfr(g)}\n{_hexfr(claim)}\n" + "".join(_hexfr(x) + "\n" for x in rc))
            f.write(f"{len(st)}\n" + "".join(f"{w} {v} {int(z)}\n" for w, v, z in st) + "".join(_hexfr(x) + "\n" for x in ch))
        res = subprocess.run([exe, "stage4", path], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        lines = [l.split() for l in res.stdout.splitlines() if l[:1] in ("E", "O")]
        rounds = [np.array([int(x, 16) for x in l[1:17]], dtype=np.uint64).reshape(4, 4) for l in lines if l[0] == "E"]
        final = np.array([int(x, 16) for x in [l for l in lines if l[0] == "O"][0][1:37]], dtype=np.uint64).reshape(9, 4)
        assert len(rounds) == 7 + log_t
        for rd in range(7 + log_t):
            we = o.computeRoundEvals(rd, claim)
            assert np.array_equal(rounds[rd], we), (k, rd)
            claim = ob.raf_update_claim(we, ch[rd])
            o.bindChallenge(rd, ch[rd])
        fc, chk = o.getFinalClaims(), o.finalCheck()
        want = [fc["val_claim"], fc["rs1_ra_claim"], fc["rs2_ra_claim"], fc["rd_wa_claim"], fc["inc_claim"], chk[0], chk[1], chk[2], claim]
        for i, w in enumerate(want):
            assert np.array_equal(final[i], w), (k, i)
        if k == 0:  # what the reference printed
            assert [ob.fr_to_int(x) for x in rounds[0]] == [le(fx["round0"]["p%d_le" % t]) for t in range(4)]
            assert ob.fr_to_int(final[8]) == le(fx["final"]["claim_le"]) == le(fx["final"]["expected_le"])
            assert ob.fr_to_int(final[5]) == le(fx["final"]["eq_scalar_le"]) and ob.fr_to_int(final[6]) == le(fx["final"]["combined_le"])


@pytest.mark.gpu
def test_cpp_streaming_outer_mirror(tmp_path, golden_dir):
    """zolt::StreamingOuterProver (compiled host code: constraint table, Lagrange weights, zg_fr_rows_affine_dev + a product session)
    against the restatement of src/zkvm/spartan/streaming_outer.zig on random cycle inputs and on the captured Stage-1 run: (t'(0), t'(inf)), the four
    evaluations of every round, and the final Az, Bz, claim and split-eq scalar, bit for bit."""
    import numpy as np
    from oracle import binding as ob
    from tests import util as U
    from tests.test_transcript_host import outer_true_claim, random_cycle_witnesses
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])
    import json
    from tests.test_transcript_host import stage1_witness_of_the_captured_run
    fx = json.load(open(os.path.join(golden_dir, "stage1_outer_rounds.json")))
    le = lambda h: int.from_bytes(bytes.fromhex(h), "little")
    for k, n in enumerate((1, 37, 512, 256)):
        if k == 3:  # the reference's captured Stage-1 run: witnesses regenerated from the ELF, its tau, r0, kernel and challenges
            w = stage1_witness_of_the_captured_run(golden_dir)
            nv = 8
            tau = np.array(fx["tau_limbs"], dtype=np.uint64)
            r0 = ob.fr_from_int(int(fx["r0_be"], 16))
            scale = ob.lagrange_kernel(r0, tau[-1])
            chals = np.stack([ob.fr_from_int(le(r["challenge_le"])) for r in fx["rounds"]])
        else:
            w = random_cycle_witnesses(300 + n, n)
            nv = max(n - 1, 0).bit_length()
            r = ob.f_to_mont(ob.FR, U.random_raw256(400 + n, 3 * nv + 8))
            tau, r0, scale, chals = r[:nv + 2], r[nv + 2], r[nv + 3], r[nv + 4:nv + 4 + nv + 1]
        o = ob.StreamingOuterProver(w, tau, scale)
        o.bindFirstRoundChallenge(r0, ob.fr_from_int(0))
        o.materializeLinearPhasePolynomials()
        o.current_claim = ob.fr_from_int(int(fx["uni_skip_claim_be"], 16)) if k == 3 else outer_true_claim(o)
        path = str(tmp_path / f"outer_{k}.txt")
        with open(path, "w") as f:
            f.write(f"{n} {nv}\n{_hexfr(scale)}\n{_hexfr(r0)}\n{_hexfr(o.current_claim)}\n")
            f.write("".join(_hexfr(x) + "\n" for x in tau) + "".join(_hexfr(x) + "\n" for x in w.reshape(-1, 4)) + "".join(_hexfr(x) + "\n" for x in chals))
        res = subprocess.run([exe, "outer", path], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        parse = lambda tag, cnt: [np.array([int(x, 16) for x in l.split()[1:1 + 4 * cnt]], dtype=np.uint64).reshape(cnt, 4)
                                  for l in res.stdout.splitlines() if l.startswith(tag + " ")]
        ts, evs, fin = parse("T", 2), parse("E", 4), parse("O", 4)[0]
        assert len(ts) == len(evs) == nv + 1
        first = ob.StreamingOuterProver(w, tau, scale)  # the UniSkip first round of the same instance
        want_s1 = first.computeFirstRoundPoly()
        assert [ob.fr_to_int(x) for x in parse("X", 9)[0]] == first.last_extended_evals, k
        assert np.array_equal(parse("S", 28)[0], want_s1), k
        for rd in range(nv + 1):
            we = o.computeRemainingRoundPoly()
            assert np.array_equal(ts[rd][0], o.last_t[0]) and np.array_equal(ts[rd][1], o.last_t[1]), (k, rd)
            assert np.array_equal(evs[rd], we), (k, rd)
            o.updateClaim(we, chals[rd])
            o.bindRemainingRoundChallenge(chals[rd])
        assert np.array_equal(fin[0], o.az[0]) and np.array_equal(fin[1], o.bz[0]) and np.array_equal(fin[2], o.current_claim)
        assert np.array_equal(fin[3], o.split_eq.current_scalar)
        if k == 3:  # what the reference printed: q(0) of every round, the claim chain, the final scalar (the first-round polynomial of the
            # compiled mirror is built with the kernel scaling here, so only the rounds are held against the log)
            assert [ob.fr_to_int(t[0]) for t in ts] == [int(r["q0_be"], 16) for r in fx["rounds"]]
            assert ob.fr_to_int(fin[3]) == le(fx["final_eq_factor_le"])


@pytest.mark.gpu
def test_cpp_original_stage4_prover_mirror(tmp_path):
    """zolt::Stage4Prover (the original, non-Gruen prover of src/zkvm/spartan/stage4_prover.zig) against the restatement: four evaluations
    and the coefficient form of every round, the final claims"""
    import numpy as np
    from oracle import binding as ob
    from tests import util as U
    from tests.test_gpu_stage4 import seeded_steps
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])
    for k, (log_t, n_steps) in enumerate(((2, 3), (7, 100), (10, 1024))):
        r = ob.f_to_mont(ob.FR, U.random_raw256(950 + log_t, 2 * log_t + 8))
        g, rc, ch, st = r[0], r[1:1 + log_t], r[1 + log_t:], seeded_steps(60 + log_t, n_steps)
        o = ob.Stage4Prover(st, g, rc)
        claim = o.computeInputClaim()
        path = str(tmp_path / f"stage4orig_{k}.txt")
        with open(path, "w") as f:
            f.write(f"{log_t} 0\n{_hexfr(g)}\n{_hexfr(claim)}\n" + "".join(_hexfr(x) + "\n" for x in rc))
            f.write(f"{len(st)}\n" + "".join(f"{w} {v} {int(z)}\n" for w, v, z in st) + "".join(_hexfr(x) + "\n" for x in ch))
        res = subprocess.run([exe, "stage4", path], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        rows = lambda tag, cnt: [np.array([int(x, 16) for x in l.split()[1:1 + 4 * cnt]], dtype=np.uint64).reshape(cnt, 4)
                                 for l in res.stdout.splitlines() if l.startswith(tag + " ")]
        evs, cos, fin = rows("E", 4), rows("P", 4), rows("O", 6)[0]
        assert len(evs) == len(cos) == 7 + log_t
        for rd in range(7 + log_t):
            we = o.computeRoundEvals(rd)
            assert np.array_equal(evs[rd], we), (k, rd)
            assert np.array_equal(cos[rd], o.computeRoundPolynomial(rd)), (k, rd)
            claim = ob.raf_update_claim(we, ch[rd])
            o.bindChallenge(rd, ch[rd])
        fc = o.getFinalClaims()
        want = [fc["val_claim"], fc["rs1_ra_claim"], fc["rs2_ra_claim"], fc["rd_wa_claim"], fc["inc_claim"], claim]
        assert all(np.array_equal(fin[i], w) for i, w in enumerate(want)), k
        assert np.array_equal(o.finalCheck()[2], claim)


@pytest.mark.gpu
def test_cpp_stage3_mirror(tmp_path, golden_dir):
    """zolt::Stage3Prover (compiled host code: ShiftPrefixSuffixProver / RegistersPrefixSuffixProver built from the witness matrix in HBM by
    zg_fr_rows_affine_dev + zg_fr_weighted_colsum_dev, the InstructionInput session, the stage's batching loop) against the restatement of
    src/zkvm/spartan/stage3_prover.zig: the captured run from the ELF (whose logged round polynomials the restatement reproduces) and
    random witnesses with odd and even variable counts — ShiftSumcheck's evaluations, the compressed polynomial, the four claims of every
    round and the final claims, bit for bit."""
    import json
    import numpy as np
    from oracle import binding as ob
    from tests import util as U
    from tests import test_transcript_host as H
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])
    P = ob._R_P
    le = lambda h: int.from_bytes(bytes.fromhex(h), "little")
    rnd = lambda seed, k: [ob.fr_to_int(x) for x in ob.f_to_mont(ob.FR, U.random_raw256(seed, k))]
    for case, n in enumerate((8, 5, 6, 2)):
        if case == 0:
            s3, w, wm, ro, rp = H.stage3_inputs_of_the_captured_run(golden_dir)
            g = int(s3["shift_gamma_be"], 16)
            ig, rg = int(s3["instr_gamma_be"], 16), int(s3["reg_gamma_be"], 16)
            claims = [le(h) for h in s3["input_claims"]]
            coeffs = [int(h, 16) for h in s3["batching_coeffs_be"]]
            ch = [le(r["challenge"]) for r in s3["rounds"]]
        else:
            T = 1 << n
            wm = ob.f_to_mont(ob.FR, U.random_raw256(7000 + n, T * 43)).reshape(T, 43, 4)
            w = [[ob.fr_to_int(x) for x in row] for row in wm]
            ro, rp, ch = rnd(7100 + n, n), rnd(7200 + n, n), rnd(7300 + n, n)
            g, ig, rg, *rest = rnd(7400 + n, 9)
            claims, coeffs = rest[:3], rest[3:]
        sg = [pow(g, i, P) for i in range(5)]
        path = str(tmp_path / f"stage3_{case}.txt")
        fi = ob.fr_from_int
        with open(path, "w") as f:
            f.write(f"{n}\n" + "".join(_hexfr(fi(x)) + "\n" for x in sg + [ig, rg] + claims + coeffs + ro + rp))
            f.write("".join(_hexfr(x) + "\n" for x in wm.reshape(-1, 4)) + "".join(_hexfr(fi(x)) + "\n" for x in ch))
        res = subprocess.run([exe, "stage3", path], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        parse = lambda tag, cnt: [[ob.fr_to_int(x) for x in np.array([int(v, 16) for v in l.split()[1:1 + 4 * cnt]], dtype=np.uint64).reshape(cnt, 4)]
                                  for l in res.stdout.splitlines() if l.startswith(tag + " ")]
        S, Pl, Cl = parse("S", 3), parse("P", 3), parse("C", 4)
        b = ob.Stage3Batch(ob.Stage3ShiftProver(w, ro, rp, sg), ob.Stage3InstructionInputProver(w, ro, rp, ig), ob.Stage3RegistersProver(w, ro, rg), claims, coeffs)
        assert len(S) == len(Pl) == len(Cl) == n
        for k in range(n):
            comp = b.computeRoundPolynomial()
            assert S[k] == b.evals[0] and Pl[k] == comp, (case, k)
            b.bindChallenge(ch[k])
            assert Cl[k] == b.claims + [b.combined], (case, k)
        sh, rgc = b.inst[0].finalClaims(), b.inst[2].finalClaims()
        assert parse("F", 5)[0] == [sh[k] for k in ("unexpanded_pc", "pc", "is_virtual", "is_first_in_sequence", "is_noop")], case
        assert parse("G", 3)[0] == [rgc[k] for k in ("rd_write_value", "rs1_value", "rs2_value")], case
        if case == 0:  # and the reference's own printed values
            assert Pl == [[le(r[c]) for c in ("c0", "c2", "c3")] for r in s3["rounds"]]
            assert [c[3] for c in Cl] == [le(r["next_claim"]) for r in s3["rounds"]]


@pytest.mark.gpu
def test_cpp_wire_formats(tmp_path, golden_dir):
    """zolt::wire (compiled host code) on the formats around the path, against the Python mirror / the oracle / the
    reference's own captured proof file: the raw SRS file (parse -> the oracle's Montgomery limbs, re-serialised byte for byte, an
    infinity record, truncation and a point off the curve), the ptau container (header, TauG1 capped by the power, AlphaTauG1, an
    infinity record, raw G2 sections, bad magic), and the ZOLT-v1 proof header (the eleven 64-byte commitments of
    logs/zolt_proof_regular.bin, re-serialised to its first 744 bytes)."""
    import numpy as np
    from oracle import binding as ob
    from oracle import pymodel as pm
    from zolt_amd import api
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])

    def run(kind, blob):
        src, dst = str(tmp_path / f"{kind}.in"), str(tmp_path / f"{kind}.out")
        open(src, "wb").write(blob)
        if os.path.exists(dst):
            os.remove(dst)
        res = subprocess.run([exe, "wire", kind, src, dst], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        return res.stdout.splitlines(), (open(dst, "rb").read() if os.path.exists(dst) else None)

    def points(lines, tag):
        rows = [l.split()[1:] for l in lines if l.startswith(tag + " ")]
        inf = [int(r[0]) for r in rows]
        xy = np.array([[int(v, 16) for v in r[1:9]] for r in rows], dtype=np.uint64).reshape(-1, 8)
        return xy, inf

    # raw SRS
    n = 33
    srs, inf = ob.hyperkzg_setup(n)
    blob = bytearray(n.to_bytes(4, "little") + b"".join(pm.commitment_bytes(p) for p in pm.mock_srs(n)) + bytes(range(64)) * 5)
    blob[4 + 64 * 5: 4 + 64 * 6] = bytes(64)  # point 5 := infinity
    lines, back = run("raw", bytes(blob))
    xy, pinf = points(lines, "P")
    want = srs.copy()
    want[5] = 0
    assert lines[0] == f"N {n}" and np.array_equal(xy, want) and pinf == [1 if i == 5 else 0 for i in range(n)]
    assert back == bytes(blob)  # trailer included
    assert run("raw", bytes(blob[:100]))[0] == ["SRSError TruncatedData"]
    bad = bytearray(blob)
    bad[4 + 64 * 3 + 63] ^= 1
    assert run("raw", bytes(bad))[0] == ["SRSError PointNotOnCurve"]
    # ptau
    hdr = (32).to_bytes(4, "little") + bytes(32) + (2).to_bytes(4, "little") + (3).to_bytes(4, "little")
    gm = ob.g1_gen_multiples(9)
    canon = ob.f_from_mont(ob.FP, gm.reshape(-1, 4)).reshape(9, 8)
    recs = [canon[i].tobytes() for i in range(9)]
    recs[3] = bytes(64)

    def ptau(sections):
        out = b"ptau" + (1).to_bytes(4, "little") + len(sections).to_bytes(4, "little")
        for typ, payload in sections:
            out += typ.to_bytes(4, "little") + len(payload).to_bytes(8, "little") + payload
        return out
    file = ptau([(1, hdr), (2, b"".join(recs)), (4, b"".join(recs[:4])), (3, bytes(128 * 5)), (6, bytes(128))])
    lines, _ = run("ptau", file)
    assert lines[0] == "H 2 3 7 4 0 640 128"
    txy, tinf = points(lines, "T")
    w7 = gm[:7].copy()
    w7[3] = 0
    assert np.array_equal(txy, w7) and tinf == [0, 0, 0, 1, 0, 0, 0]
    axy, ainf = points(lines, "A")
    assert np.array_equal(axy[:3], gm[:3]) and ainf == [0, 0, 0, 1]
    got = api.srs_g1_from_ptau(file)  # the Python mirror reads the same file the same way
    assert np.array_equal(got["powers_of_tau_g1"][0], txy) and got["power"] == 2 and got["ceremony_power"] == 3
    assert run("ptau", bytes(24))[0] == ["SRSError InvalidFileFormat"]
    assert run("ptau", b"ptau")[0] == ["SRSError TruncatedData"]
    # the reference's captured proof
    proof = open(os.path.join(golden_dir, "zolt_proof_regular.bin"), "rb").read()
    lines, header = run("proof", proof)
    want_c = api.parse_zolt_proof_commitments(proof)
    got_c = {l.split()[1]: bytes.fromhex(l.split()[2]) for l in lines if l.startswith("C ")}
    assert got_c == want_c and header == proof[:744]
    g = ob.g1_gen_multiples(1)[0]
    assert bytes.fromhex([l for l in lines if l.startswith("G ")][0].split()[1]) == api.commitment_to_bytes(g, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n_evals,sigma,nu", [(64, 3, 3), (61, 3, 3), (1000, 5, 5), (5, 4, 2)])
def test_dory_row_commitments_and_vector_matrix_product(tmp_path, n_evals, sigma, nu):
    """Dory's two data-parallel pieces (src/poly/commitment/dory.zig:622-670) — row commitments = a batch of MSMs over a prefix of g1_vec
    (ragged last row, fewer evaluations than one row), vector-matrix product = a weighted column sum (left_vec shorter than the row count,
    evaluations shorter than the matrix) — through the Python mirror and the compiled mirror, against the restatement, bit for bit."""
    import numpy as np
    from oracle import binding as ob
    from tests import util as U
    from zolt_amd import api, lib
    lib.init()
    cols = 1 << sigma
    g1 = ob.g1_gen_multiples(cols)
    g1_inf = np.zeros(cols, dtype=np.uint8)
    g1_inf[2] = 1  # an identity element inside g1_vec
    ev = ob.f_to_mont(ob.FR, U.random_raw256(6000 + n_evals, n_evals))
    ev[3] = 0
    left = ob.f_to_mont(ob.FR, U.random_raw256(6100 + n_evals, max(1, (1 << nu) - 1)))  # one short of the row count
    want_rc, want_inf = ob.dory_row_commitments(g1, g1_inf, ev, cols)
    want_v = ob.dory_vector_matrix_product(ev, left, nu, sigma)
    b = lib.Bases.upload(g1, g1_inf)
    try:
        rc, inf = api.Dory.computeRowCommitments(b, ev, cols)
    finally:
        b.free()
    assert np.array_equal(inf, want_inf) and np.array_equal(rc, want_rc)
    assert np.array_equal(api.Dory.computeVectorMatrixProduct(ev, left, nu, sigma), want_v)
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])
    path = str(tmp_path / "dory.txt")
    with open(path, "w") as f:
        f.write(f"{cols}\n" + "".join(" ".join("%x" % int(v) for v in g1[i]) + f" {int(g1_inf[i])}\n" for i in range(cols)))
        f.write(f"{cols} {n_evals}\n" + "".join(_hexfr(x) + "\n" for x in ev))
        f.write(f"{nu} {sigma} {left.shape[0]}\n" + "".join(_hexfr(x) + "\n" for x in left))
    res = subprocess.run([exe, "dory", path], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    rows = [l.split()[1:] for l in res.stdout.splitlines() if l.startswith("R ")]
    assert [int(r[0]) for r in rows] == list(want_inf)
    got = np.array([[int(v, 16) for v in r[1:9]] for r in rows], dtype=np.uint64).reshape(-1, 8)
    assert np.array_equal(got[want_inf == 0], want_rc[want_inf == 0])
    v = [l for l in res.stdout.splitlines() if l.startswith("V")][0].split()[1:]
    assert np.array_equal(np.array([int(x, 16) for x in v], dtype=np.uint64).reshape(-1, 4), want_v)


@pytest.mark.gpu
@pytest.mark.parametrize("d,nu,sigma", [(0, 2, 2), (2, 3, 3), (3, 3, 3), (6, 3, 3), (8, 3, 3), (5, 2, 4)])
def test_dory_evaluation_vectors(d, nu, sigma):
    """multilinearLagrangeBasis / computeEvaluationVectors (dory.zig:544-620) on the device's eq-table kernel (reversed point) against the
    restatement's product formula: fewer variables than columns, an exact split, more variables than the matrix holds"""
    import numpy as np
    from oracle import binding as ob
    from tests import util as U
    from zolt_amd import api, lib
    lib.init()
    pt = ob.f_to_mont(ob.FR, U.random_raw256(6200 + d, max(d, 1)))[:d]
    wl, wr = ob.dory_evaluation_vectors(pt, nu, sigma)
    gl, gr = api.Dory.computeEvaluationVectors(pt, nu, sigma)
    assert np.array_equal(gl, wl) and np.array_equal(gr, wr)


@pytest.mark.gpu
def test_cpp_mirror_produces_the_captured_proof_file(golden_dir, tmp_path):
    """BASELINE config 5 from COMPILED host code (the stand-in for the patched Zig modules): zolt::Transcript, zolt::LassoProver,
    zolt::proveStage4 / 5 / 6 and the device folds produce the stage records of the reference's captured proof file; re-serialised
    (src/zkvm/serialization.zig:186-343) they are bytes 744 .. 11345 of tests/golden/zolt_proof_regular.bin. Inputs regenerated from the
    ELF by tests/util.py (trace, lookup indices) and the oracle's R1CS restatement (Az, Bz), as in tests/test_transcript_host.py."""
    import numpy as np
    from oracle import binding as ob  # fixture regeneration (the R1CS witness of the captured run) and representation conversions
    from tests import util as U
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])
    P = U.proof_file_sections()
    data = open(os.path.join(golden_dir, "zolt_proof_regular.bin"), "rb").read()
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    w = ob.r1cs_witness_from_trace(U.fibonacci_full_trace(elf))
    n = w.shape[0]
    az = np.zeros((1 << 13, 4), dtype=np.uint64)
    bz = np.zeros((1 << 13, 4), dtype=np.uint64)
    for i, (cond, left, right) in enumerate(ob.UNIFORM_CONSTRAINTS):
        az[np.arange(n) * 19 + i] = ob._lc_eval(cond, w)
        bz[np.arange(n) * 19 + i] = ob._fsub(ob._lc_eval(left, w), ob._lc_eval(right, w))
    idx = U.fibonacci_lookup_indices(elf)
    steps = U.fibonacci_trace_steps(elf)
    path = tmp_path / "proof_case.txt"
    with open(path, "w") as f:
        for off in P["absorbed_commitment_offsets"].values():
            f.write(data[off:off + 64].hex() + "\n")
        f.write(f"{P['log_t']} {P['log_k']} {len(steps)}\n" + " ".join(str(wd) for wd, _, _ in steps) + "\n")
        f.write(f"{idx.shape[0]}\n" + "".join(f"{int(lo)} {int(hi)}\n" for lo, hi in idx))
        f.write(f"{az.shape[0]}\n")
        for t in (az, bz):
            f.write("".join(_hexfr(x) + "\n" for x in t))
    res = subprocess.run([exe, "proof", str(path)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    stages, cur = [], None
    to_int = lambda words: ob.fr_to_int(np.array([int(x, 16) for x in words], dtype=np.uint64))
    for line in res.stdout.splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "S":
            cur = ([], [], [])
            stages.append(cur)
        elif t[0] == "P":
            cur[0].append([to_int(t[1 + 4 * k:5 + 4 * k]) for k in range((len(t) - 1) // 4)])
        elif t[0] == "H":
            cur[1].extend(to_int(t[1 + 4 * k:5 + 4 * k]) for k in range((len(t) - 1) // 4))
        elif t[0] == "C":
            cur[2].extend(to_int(t[1 + 4 * k:5 + 4 * k]) for k in range((len(t) - 1) // 4))
    assert len(stages) == 6
    assert U.serialize_stage_sections(P["log_t"], P["log_k"], stages) == data[744:]


@pytest.mark.gpu
def test_cpp_witness_matrix_from_trace_columns(tmp_path, golden_dir):
    """zolt::CycleColumns::fromTrace (compiled host code: the integer-domain restatement of R1CSWitnessGenerator.generateWitness) +
    zg_fr_rows_from_columns against the oracle's restatement of the same generator: every element of the device-widened matrix, on the
    captured fibonacci run and on random traces that reach every opcode branch; StreamingOuterProver reads the shared matrix and ready
    rows alike."""
    import numpy as np
    from oracle import binding as ob
    from tests import util as U
    from tests.test_witness_columns import random_trace
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    for k, steps in enumerate((U.fibonacci_full_trace(elf), random_trace(21, 600, 40), random_trace(22, 33, 0))):
        path = str(tmp_path / f"trace_{k}.txt")
        with open(path, "w") as f:
            f.write(f"{len(steps)}\n")
            for st in steps:
                f.write("%x %x %x %x %x %x %d %x %d %d\n" % (st["instruction"], st["pc"], st["unexpanded_pc"], st["rs1_value"], st["rs2_value"], st["rd_value"],
                                                            st["memory_value"] is not None, st["memory_value"] or 0, st["is_compressed"], st["is_noop"]))
        res = subprocess.run([exe, "witness", path], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        rows = [np.array([int(x, 16) for x in l.split()[1:]], dtype=np.uint64).reshape(43, 4) for l in res.stdout.splitlines() if l.startswith("W ")]
        assert np.array_equal(np.stack(rows), ob.r1cs_witness_from_trace(steps)), k
        s1 = [l for l in res.stdout.splitlines() if l.startswith("S ")]
        assert len(s1) == 2 and s1[0] == s1[1]
        bpc = int([l for l in res.stdout.splitlines() if l.startswith("B ")][0].split()[1])
        assert bpc == 156
        assert "T 1" in res.stdout.splitlines(), "CycleWitnessMatrix::fromTrace in slices differs from the one-call matrix"


@pytest.mark.gpu
def test_prove_path_composite_regenerates_the_captured_proof_file(golden_dir, tmp_path):
    """tools/bench_prove_path — ONE sequence from compiled host code: HyperKZG.setup at the reference's srs_size, the three commitments from
    machine words, the witness matrix widened on the device from integer trace columns, Az / Bz materialised on the device in JoltR1CS's
    layout, the fused eq * Az * Bz open and its thirteen LowToHigh rounds, RAF, Lasso, stages 4-6 — fed with nothing but what the ELF
    yields (trace steps, lookup indices, program bytes). Its header commitments and stage records re-serialised ARE the reference's
    captured proof file: all 11 345 bytes (BASELINE config 5), and the same run prints the per-call cost breakdown of the sequence."""
    import hashlib
    import json
    import numpy as np
    from oracle import binding as ob  # representation conversions only
    from tests import util as U
    from zolt_amd import api
    exe = os.path.join(ROOT, "tools", "bench_prove_path")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "../../tools/bench_prove_path"])
    P = U.proof_file_sections()
    data = open(os.path.join(golden_dir, "zolt_proof_regular.bin"), "rb").read()
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    steps = U.fibonacci_full_trace(elf)
    idx = U.fibonacci_lookup_indices(elf)
    code = elf[0x1000:0x1000 + 104]
    path = tmp_path / "prove_case.txt"
    with open(path, "w") as f:
        f.write(f"{P['log_t']} {P['log_k']} 1280 80000000\n{len(steps)}\n")
        for st in steps:
            f.write("%x %x %x %x %x %x %d %x %d %d\n" % (st["instruction"], st["pc"], st["unexpanded_pc"], st["rs1_value"], st["rs2_value"], st["rd_value"],
                                                        st["memory_value"] is not None, st["memory_value"] or 0, st["is_compressed"], st["is_noop"]))
        f.write("0\n")  # the fibonacci run touches no RAM
        f.write(f"{idx.shape[0]}\n" + "".join(f"{int(lo)} {int(hi)}\n" for lo, hi in idx))
        f.write(f"{len(code)}\n" + code.hex() + "\n")
    res = subprocess.run([exe, "file", str(path), "1"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    # a key planned for one use (no table of multiples: zg_msm_config.expected_uses = 1) proves the same bytes
    res1 = subprocess.run([exe, "file", str(path), "1", "1"], capture_output=True, text=True, timeout=600)
    assert res1.returncode == 0, res1.stdout[-2000:] + res1.stderr[-2000:]
    records = lambda out: [l for l in out.splitlines() if l[:2] in ("K ", "S ", "P ", "H ", "C ")]
    assert records(res1.stdout) == records(res.stdout) and len(records(res.stdout)) > 50
    stages, cur, comm = [], None, {}
    to_int = lambda words: ob.fr_to_int(np.array([int(x, 16) for x in words], dtype=np.uint64))
    for line in res.stdout.splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "K":
            comm[t[1]] = bytes.fromhex(t[2])
        elif t[0] == "S":
            cur = ([], [], [])
            stages.append(cur)
        elif t[0] == "P":
            cur[0].append([to_int(t[1 + 4 * k:5 + 4 * k]) for k in range((len(t) - 1) // 4)])
        elif t[0] == "H":
            cur[1].extend(to_int(t[1 + 4 * k:5 + 4 * k]) for k in range((len(t) - 1) // 4))
        elif t[0] == "C":
            cur[2].extend(to_int(t[1 + 4 * k:5 + 4 * k]) for k in range((len(t) - 1) // 4))
    assert len(stages) == 6 and set(comm) == {"bytecode", "memory", "register"}
    want = api.parse_zolt_proof_commitments(data)
    assert comm["bytecode"] == want["bytecode.commitment"] and comm["memory"] == want["memory.commitment"] == bytes(64) and comm["register"] == want["register.commitment"]
    # the header a `zolt prove` run writes from these three commitments (every other commitment slot of this run is the identity)
    offs = P["absorbed_commitment_offsets"]
    composed = bytearray(data[:744])
    for name, key in (("bytecode", "bytecode"), ("memory", "memory"), ("register", "registers")):
        composed[offs[key]:offs[key] + 64] = bytes(64)
        composed[offs[key]:offs[key] + 64] = comm[name]
    tail = U.serialize_stage_sections(P["log_t"], P["log_k"], stages)
    assert tail == data[744:]
    assert hashlib.sha256(bytes(composed) + tail).digest() == hashlib.sha256(data).digest() and len(composed) + len(tail) == 11345
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])["prove_path"]
    assert line["log_t"] == 8 and len(line["steps"]) >= 15 and len(line["top3"]) == 3 and line["total_ms"] > 0
