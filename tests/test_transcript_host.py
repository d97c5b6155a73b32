"""The host Keccak transcript (src/transcripts/mod.zig:49-221) — CPU only. Keccak-f[1600] of the C oracle and of the big-int
model is pinned against an INDEPENDENT implementation (hashlib's SHA3-256 uses the same permutation); the transcript wrappers
(oracle C, oracle/pymodel.py, zolt_amd/api.py — the product's host mirror) must then agree byte for byte on every absorb /
squeeze sequence, including the rate boundary at 136 bytes."""
import hashlib

import numpy as np

from oracle import binding as ob
from oracle import pymodel as pm
from tests import util as U


def _sha3_256_with(perm, msg):
    st = bytearray(200)
    m = bytearray(msg)
    m.append(0x06)
    while len(m) % 136:
        m.append(0)
    m[-1] |= 0x80
    for off in range(0, len(m), 136):
        for i in range(136):
            st[i] ^= m[off + i]
        lanes = perm([int.from_bytes(st[8 * i:8 * i + 8], "little") for i in range(25)])
        for i, v in enumerate(lanes):
            st[8 * i:8 * i + 8] = int(v).to_bytes(8, "little")
    return bytes(st[:32])


def test_keccak_f_pinned_against_hashlib_sha3():
    for msg in (b"", b"abc", bytes(range(200)), b"x" * 135, b"y" * 136, b"z" * 137, b"Jolt" * 100):
        want = hashlib.sha3_256(msg).digest()
        assert _sha3_256_with(pm.keccak_f1600, msg) == want
        assert _sha3_256_with(lambda l: [int(x) for x in ob.keccak_f1600(l)], msg) == want


def test_transcript_mirrors_agree():
    from zolt_amd import api
    rng = np.random.default_rng(11)
    for domain in (b"Jolt", b"", b"d" * 136, b"e" * 300):
        a, b, c = api.Transcript(domain), ob.Transcript(domain), pm.KeccakTranscript(domain)
        for k in range(60):
            op = int(rng.integers(0, 3))
            if op == 0:
                data = bytes(rng.integers(0, 256, size=int(rng.integers(0, 300)), dtype=np.uint8))
                a.appendBytes(data); b.append_bytes(data); c.append_bytes(data)
            elif op == 1:
                s = ob.f_to_mont(ob.FR, U.random_raw256(100 + k, 1))[0]
                a.appendScalar(b"lbl%d" % k, s); b.append_scalar(b"lbl%d" % k, s); c.append_scalar_mont(b"lbl%d" % k, pm.from_limbs(s))
            else:
                x, y, z = a.challengeScalar(b"ch"), b.challenge_scalar(b"ch"), c.challenge_scalar(b"ch")
                assert np.array_equal(x, y) and U.fr_to_int(x) == z
                assert pm.from_limbs(x) < pm.R_MOD  # canonical Montgomery limbs
            sb, pos = b.state_bytes()
            assert bytes(a.state) == sb == bytes(c.state) and a.position == pos == c.position
    # the reference's convention (:100-110): appendScalar = label bytes then the 32 little-endian bytes of the Montgomery limbs
    t1, t2 = api.Transcript(b"Jolt"), api.Transcript(b"Jolt")
    s = ob.f_from_u64(ob.FR, np.array([7], dtype=np.uint64))[0]
    t1.appendScalar(b"round_poly_0", s)
    t2.appendBytes(b"round_poly_0" + s.astype("<u8").tobytes())
    assert t1.state == t2.state


def test_stage1_oracle_kat_tiny():
    """Stage-1 loop restated in the oracle on [1,2,3,4] with a fresh transcript: p0 = 1 + 3, p1 = 2 + 4, p2 = 2 p1 - p0; after the
    fold by r the next p0 + p1 equals p(r) = p0 + r (p1 - p0) (the sumcheck invariant the reference prints as sumcheck_ok)."""
    poly = U.fr([1, 2, 3, 4])
    rp, ch, fin = ob.stage1_prove(poly, 2, ob.Transcript(b"Jolt"))
    assert [U.fr_to_int(x) for x in rp[0]] == [4, 6, 8]
    r0 = U.fr_to_int(ch[0])
    assert (U.fr_to_int(rp[1][0]) + U.fr_to_int(rp[1][1])) % pm.R_MOD == (4 + r0 * 2) % pm.R_MOD
    r1 = U.fr_to_int(ch[1])
    assert U.fr_to_int(fin) == (U.fr_to_int(rp[1][0]) + r1 * (U.fr_to_int(rp[1][1]) - U.fr_to_int(rp[1][0]))) % pm.R_MOD
    # more rounds than variables: [poly[0], 0, 0] rounds, no further folding (jolt_r1cs.zig:421-430,462-465)
    rp3, ch3, fin3 = ob.stage1_prove(poly, 4, ob.Transcript(b"Jolt"))
    assert np.array_equal(rp3[:2], rp) and np.array_equal(rp3[2][0], fin) and not rp3[2][1:].any() and np.array_equal(fin3, fin)


def test_blake2b_transcript_matches_reference_log(golden_dir):
    """The Jolt-compatible transcript of the reference's proving path (src/transcripts/blake2b.zig) against the states the
    reference printed in its captured run (logs/zolt.log:28-30,1165-1187; fixture tests/golden/blake2b_transcript_preamble.json):
    the full 32-byte state after init("Jolt"), then six appendU64 and two empty appendBytes whose 8-byte state prefixes the log
    shows, up to the state the first appendGT saw."""
    import json
    import os
    from zolt_amd import api
    d = json.load(open(os.path.join(golden_dir, "blake2b_transcript_preamble.json")))
    t = api.Blake2bTranscript(d["label"].encode())
    assert t.state.hex() == d["initial_state_hex"] and t.n_rounds == 0
    for op in d["ops"]:
        if op["op"] == "appendU64":
            t.appendU64(op["value"])
        else:
            assert t.state.hex().startswith(op["state_before_prefix"])
            t.appendBytes(bytes.fromhex(op["hex"]))
            assert t.state.hex().startswith(op["state_after_prefix"])
    assert t.state.hex().startswith(d["state_prefix_after_all_ops"]) and t.n_rounds == len(d["ops"])


def test_blake2b_transcript_challenge_shapes():
    """challengeScalar = MontU128Challenge: raw limbs [0, 0, low, high] of a 125-bit value (the layout the reference's
    r_cycle_be fixture shows); challengeScalarFull = the unmasked 128-bit value in Montgomery form; both advance the state."""
    from zolt_amd import api
    t = api.Blake2bTranscript(b"Jolt")
    t.appendScalar(api.fr_from_int(12345))
    t.appendScalars([api.fr_from_int(1), api.fr_from_int(2)])
    t2 = api.Blake2bTranscript(b"Jolt")
    t2.appendBytes((12345).to_bytes(32, "big"))
    t2.appendMessage(b"begin_append_vector")
    t2.appendBytes((1).to_bytes(32, "big"))
    t2.appendBytes((2).to_bytes(32, "big"))
    t2.appendMessage(b"end_append_vector")
    assert t.state == t2.state and t.n_rounds == 5
    t3 = api.Blake2bTranscript(b"Jolt")
    t3.state, t3.n_rounds = t.state, t.n_rounds
    raw = t3.challengeBytes(16)  # the 16 bytes all three draw from the same state
    c = t.challengeScalar()
    # challengeScalar128Bits (:332-390): bytes reversed, read big-endian (= the raw bytes little-endian), 125-bit mask, limbs UNCONVERTED
    want = int.from_bytes(raw, "little") & ((1 << 125) - 1)
    assert c[0] == 0 and c[1] == 0 and int(c[2]) | (int(c[3]) << 64) == want and int(c[3]) < (1 << 61)
    # challengeScalarFull (:279-312): bytes reversed, read little-endian (= the raw bytes big-endian), no mask, proper Montgomery form
    full = t2.challengeScalarFull()
    assert api.fr_to_int(full) == int.from_bytes(raw, "big")
    assert t.state == t2.state == t3.state
    assert len(t.challengeBytes(70)) == 70 and t.n_rounds == t2.n_rounds + 3
    v = t2.challengeVector(3)  # :392-399: challengeScalar each
    assert v.shape == (3, 4) and not v[:, :2].any()
