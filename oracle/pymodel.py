"""Independent Python big-int model of the Zolt hot path (TEST INFRASTRUCTURE).

Plain `int` arithmetic, affine group law, no Montgomery tricks: a second,
structurally different statement of the same mathematics as
oracle/zolt_oracle.c, used to cross-check it and to generate golden vectors
(tests/golden/). Only tests/ and the golden generator import this.

Reference semantics being modelled (paths under /root/reference):
  field constants          src/field/mod.zig:16-41 (Fr), :51-75 (Fp)
  curve  y^2 = x^3 + 3     src/msm/mod.zig:11-12,106-115; G = (1,2) :43-49
  MSM.compute = Σ sᵢ·Pᵢ    src/msm/mod.zig:355-438 (result is the unique affine point)
  eq table (big-endian)    src/poly/mod.zig:252-290
  bindFirst / bindLow      src/poly/mod.zig:128-175
  runSumcheck + toy mixer  src/subprotocols/mod.zig:69-122,165-243,302-354
  mock SRS, commit         src/poly/commitment/mod.zig:174-255
  64-byte BE commitment    src/zkvm/commitment_types.zig:49-54
"""
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr
P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fp
MONT_R = 1 << 256
M64 = (1 << 64) - 1
G1 = (1, 2)
TAU = 0x12345678


def to_mont(v, mod):
    return (v % mod) * MONT_R % mod


def from_mont(v, mod):
    return v * pow(MONT_R, -1, mod) % mod


def limbs(v):
    return [(v >> (64 * i)) & M64 for i in range(4)]


def from_limbs(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


# ---- affine group law; None is the point at infinity
def ec_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % P_MOD == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P_MOD) % P_MOD
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P_MOD) % P_MOD
    x3 = (lam * lam - x1 - x2) % P_MOD
    return (x3, (lam * (x1 - x3) - y1) % P_MOD)


def ec_neg(p):
    return None if p is None else (p[0], (-p[1]) % P_MOD)


def ec_mul(k, p):
    k %= R_MOD
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, p)
        p = ec_add(p, p)
        k >>= 1
    return acc


def on_curve(p):
    return p is None or (p[1] * p[1] - p[0] ** 3 - 3) % P_MOD == 0


def msm(points, scalars):
    """Σ sᵢ·Pᵢ with canonical integer scalars; points are affine tuples or None."""
    acc = None
    for p, s in zip(points, scalars):
        acc = ec_add(acc, ec_mul(s, p))
    return acc


def msm_generator_multiples(ks, scalars):
    """Closed form for bases Pᵢ = kᵢ·G: (Σ sᵢ·kᵢ mod r)·G."""
    return ec_mul(sum(k * s for k, s in zip(ks, scalars)) % R_MOD, G1)


def commitment_bytes(p):
    if p is None:
        return bytes(64)
    return p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")


def mock_srs(n):
    return [ec_mul(pow(TAU, i, R_MOD), G1) for i in range(n)]


# ---- poly / sumcheck over Fr (canonical ints)
def eq_table(r):
    """EqPolynomial.evals: index MSB <-> r[0]."""
    out = [1]
    for rj in r:  # appending a variable as the new LSB == the reference's reverse doubling
        out = [v for x in out for v in ((x * (1 - rj)) % R_MOD, (x * rj) % R_MOD)]
    return out


def bind_high(t, r):
    h = len(t) // 2
    return [((1 - r) * t[i] + r * t[i + h]) % R_MOD for i in range(h)]


def bind_low(t, r):
    return [(t[2 * i] + r * (t[2 * i + 1] - t[2 * i])) % R_MOD for i in range(len(t) // 2)]


def derive_challenge(rnd, claim, coeffs):
    """Toy mixer on raw MONTGOMERY limbs — src/subprotocols/mod.zig:211-243."""
    h = 0x9E3779B97F4A7C15
    h ^= rnd
    h = (h * 0xFF51AFD7ED558CCD) & M64
    for l in limbs(to_mont(claim, R_MOD)):
        h ^= l
        h = (h * 0xC4CEB9FE1A85EC53) & M64
    for c in coeffs:
        for l in limbs(to_mont(c, R_MOD)):
            h ^= l
            h = (h * 0xFF51AFD7ED558CCD) & M64
            h ^= h >> 33
    h ^= h >> 33
    h = (h * 0xFF51AFD7ED558CCD) & M64
    h ^= h >> 33
    return h  # F.fromU64(h): canonical value h


def run_sumcheck(evals):
    t = [e % R_MOD for e in evals]
    claim = sum(t) % R_MOD
    vclaim = claim
    rounds, chals = [], []
    rnd = 0
    while len(t) > 1:
        h = len(t) // 2
        g0, g1 = sum(t[:h]) % R_MOD, sum(t[h:]) % R_MOD
        coeffs = [g0, (g1 - g0) % R_MOD]
        assert (coeffs[0] + coeffs[0] + coeffs[1]) % R_MOD == vclaim
        c = derive_challenge(rnd, vclaim, coeffs)
        vclaim = (coeffs[0] + coeffs[1] * c) % R_MOD
        t = bind_high(t, c)
        rounds.append(coeffs)
        chals.append(c)
        rnd += 1
    return claim, rounds, chals, t[0], vclaim == t[0]


# ---- Keccak Fiat-Shamir transcript (src/transcripts/mod.zig:19-221), independent big-int / byte model
_KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b, 0x0000000080000001,
              0x8000000080008081, 0x8000000000008009, 0x000000000000008a, 0x0000000000000088, 0x0000000080008009, 0x000000008000000a,
              0x000000008000808b, 0x800000000000008b, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
              0x000000000000800a, 0x800000008000000a, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_KECCAK_ROTC = [1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44]
_KECCAK_PILN = [10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1]
_M64 = (1 << 64) - 1


def keccak_f1600(st):
    """st: list of 25 u64 lanes -> permuted lanes (24 rounds, src/transcripts/mod.zig:163-213)."""
    st = list(st)
    rotl = lambda x, n: ((x << n) | (x >> (64 - n))) & _M64
    for rnd in range(24):
        bc = [st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20] for i in range(5)]
        for i in range(5):
            t = bc[(i + 4) % 5] ^ rotl(bc[(i + 1) % 5], 1)
            for j in range(i, 25, 5):
                st[j] ^= t
        t = st[1]
        for i in range(24):
            j = _KECCAK_PILN[i]
            st[j], t = rotl(t, _KECCAK_ROTC[i]), st[j]
        for row in range(0, 25, 5):
            b = st[row:row + 5]
            for i in range(5):
                st[row + i] = b[i] ^ (~b[(i + 1) % 5] & _M64 & b[(i + 2) % 5])
        st[0] ^= _KECCAK_RC[rnd]
    return st


class KeccakTranscript:
    """Transcript(F) of the reference: bytes are XORed into a 200-byte state at `position`, a Keccak-f every 136 bytes (no
    padding, no squeeze offset); challengeScalar = append label, one Keccak-f, F.fromBytes(state[0..32]) (little-endian, reduced
    mod r). Field elements are absorbed as their raw MONTGOMERY limbs, little-endian (appendScalar, :100-110)."""
    RATE = 136

    def __init__(self, domain=b"Jolt"):
        self.state = bytearray(200)
        self.position = 0
        self.append_bytes(domain)

    def _permute(self):
        lanes = [int.from_bytes(self.state[8 * i:8 * i + 8], "little") for i in range(25)]
        lanes = keccak_f1600(lanes)
        for i, v in enumerate(lanes):
            self.state[8 * i:8 * i + 8] = v.to_bytes(8, "little")

    def append_bytes(self, data):
        for byte in bytes(data):
            self.state[self.position] ^= byte
            self.position += 1
            if self.position >= self.RATE:
                self._permute()
                self.position = 0

    def append_scalar_mont(self, label, mont_value):
        """mont_value: the element's Montgomery representative as an int (what scalar.limbs hold)"""
        self.append_bytes(label)
        self.append_bytes(int(mont_value).to_bytes(32, "little"))

    def challenge_scalar(self, label, mod=R_MOD):
        """-> canonical int value of the challenge (F.fromBytes reduces the 256-bit little-endian integer mod r)"""
        self.append_bytes(label)
        self._permute()
        return int.from_bytes(self.state[:32], "little") % mod
