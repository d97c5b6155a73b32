// sumcheck.hpp — Keccak / Blake2b transcripts (host, between rounds), Sumcheck prover / verifier, runSumcheck.
// Part of zolt_host.hpp (the C++ host mirror over include/zolt_gpu.h); included by it, after the parts it depends on.
#pragma once
#ifndef ZOLT_HOST_UMBRELLA
#error "include zolt_host.hpp"
#endif
namespace zolt {

// ---------------------------------------------------------------- sumcheck
// ---------------------------------------------------------------- transcript (host side, between rounds)
// Transcript(F) — the reference's Keccak Fiat-Shamir transcript, src/transcripts/mod.zig:49-221: bytes XORed into a 200-byte
// state at `position`, Keccak-f[1600] every 136 bytes, challengeScalar = label, one Keccak-f, F.fromBytes(state[0..32]).
class Transcript {
public:
    explicit Transcript(const std::string &domain = "Jolt") { appendBytes(reinterpret_cast<const uint8_t *>(domain.data()), domain.size()); }
    void appendBytes(const uint8_t *data, size_t n) {  // :88-98
        for (size_t i = 0; i < n; i++) {
            state_[position_] ^= data[i];
            position_ += 1;
            if (position_ >= 136) {
                keccakF();
                position_ = 0;
            }
        }
    }
    void appendBytes(const std::string &s) { appendBytes(reinterpret_cast<const uint8_t *>(s.data()), s.size()); }
    void appendScalar(const std::string &label, const Fr &scalar) {  // :100-110: raw Montgomery limbs, little-endian
        appendBytes(label);
        uint8_t buf[32];
        for (int i = 0; i < 4; i++)
            for (int b = 0; b < 8; b++) buf[8 * i + b] = (uint8_t)(scalar.limbs[i] >> (8 * b));
        appendBytes(buf, 32);
    }
    Fr challengeScalar(const std::string &label) {  // :116-130
        appendBytes(label);
        keccakF();
        return Fr::fromBytes(state_);
    }
    const uint8_t *state() const { return state_; }

private:
    uint8_t state_[200] = {0};
    size_t position_ = 0;
    static uint64_t rotl(uint64_t x, unsigned n) { return (x << n) | (x >> (64 - n)); }
    void keccakF() {  // :163-213
        static const uint64_t RC[24] = {
            0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
            0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
            0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
            0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        static const unsigned ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
        static const unsigned PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
        uint64_t st[25];
        for (int i = 0; i < 25; i++) {
            uint64_t v = 0;
            for (int b = 7; b >= 0; b--) v = (v << 8) | state_[8 * i + b];
            st[i] = v;
        }
        for (int round = 0; round < 24; round++) {
            uint64_t bc[5];
            for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
            for (int i = 0; i < 5; i++) {
                uint64_t t = bc[(i + 4) % 5] ^ rotl(bc[(i + 1) % 5], 1);
                for (int j = i; j < 25; j += 5) st[j] ^= t;
            }
            uint64_t t = st[1];
            for (int i = 0; i < 24; i++) {
                unsigned j = PILN[i];
                uint64_t tmp = st[j];
                st[j] = rotl(t, ROTC[i]);
                t = tmp;
            }
            for (int row = 0; row < 25; row += 5) {
                for (int i = 0; i < 5; i++) bc[i] = st[row + i];
                for (int i = 0; i < 5; i++) st[row + i] = bc[i] ^ (~bc[(i + 1) % 5] & bc[(i + 2) % 5]);
            }
            st[0] ^= RC[round];
        }
        for (int i = 0; i < 25; i++)
            for (int b = 0; b < 8; b++) state_[8 * i + b] = (uint8_t)(st[i] >> (8 * b));
    }
};

// Blake2b-256 (RFC 7693, unkeyed) for the Jolt-compatible transcript
inline void blake2b256(const uint8_t *in, size_t inlen, uint8_t out[32]) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    static const uint8_t SIGMA[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t h[8];
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010000ULL ^ 32ULL;
    auto rotr = [](uint64_t x, unsigned n) { return (x >> n) | (x << (64 - n)); };
    auto compress = [&](const uint8_t *block, uint64_t t, bool last) {
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; i++) {
            uint64_t w = 0;
            for (int b = 7; b >= 0; b--) w = (w << 8) | block[8 * i + b];
            m[i] = w;
        }
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[i + 8] = IV[i];
        }
        v[12] ^= t;
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t *sg = SIGMA[r];
            G(0, 4, 8, 12, m[sg[0]], m[sg[1]]);   G(1, 5, 9, 13, m[sg[2]], m[sg[3]]);
            G(2, 6, 10, 14, m[sg[4]], m[sg[5]]);  G(3, 7, 11, 15, m[sg[6]], m[sg[7]]);
            G(0, 5, 10, 15, m[sg[8]], m[sg[9]]);  G(1, 6, 11, 12, m[sg[10]], m[sg[11]]);
            G(2, 7, 8, 13, m[sg[12]], m[sg[13]]); G(3, 4, 9, 14, m[sg[14]], m[sg[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    };
    size_t off = 0;
    while (inlen - off > 128) {
        compress(in + off, off + 128, false);
        off += 128;
    }
    uint8_t blk[128] = {0};
    std::memcpy(blk, in + off, inlen - off);
    compress(blk, inlen, true);
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(h[i] >> (8 * b));
}

// Blake2bTranscript(F) — the Jolt-compatible transcript of the reference's proving path (src/transcripts/blake2b.zig:25-545): a 32-byte
// state and a round counter; every operation hashes state || [0u8; 28] || n_rounds_be32 || payload, the digest is the new state.
class Blake2bTranscript {
public:
    uint8_t state[32];
    uint32_t n_rounds = 0;
    explicit Blake2bTranscript(const std::string &label = "Jolt") {  // :39-69
        uint8_t padded[32] = {0};
        std::memcpy(padded, label.data(), label.size() < 32 ? label.size() : 32);
        blake2b256(padded, 32, state);
    }
    void appendMessage(const std::string &msg) {  // :96-120: right-padded to 32 bytes
        uint8_t padded[32] = {0};
        std::memcpy(padded, msg.data(), msg.size() < 32 ? msg.size() : 32);
        hashWith(padded, 32, nullptr);
    }
    void appendBytes(const uint8_t *data, size_t n) { hashWith(data, n, nullptr); }  // :123-156
    void appendU64(uint64_t x) {  // :160-176: [0u8; 24] ++ x.to_be_bytes()
        uint8_t buf[32] = {0};
        for (int b = 0; b < 8; b++) buf[24 + b] = (uint8_t)(x >> (8 * (7 - b)));
        hashWith(buf, 32, nullptr);
    }
    void appendScalar(const Fr &scalar) {  // :182-200: the canonical value, big-endian
        Fr one_raw{{1, 0, 0, 0}};
        Fr canon = scalar.mul(one_raw);  // fromMontgomery
        uint8_t buf[32];
        for (int i = 0; i < 4; i++)
            for (int b = 0; b < 8; b++) buf[31 - (8 * i + b)] = (uint8_t)(canon.limbs[i] >> (8 * b));
        hashWith(buf, 32, nullptr);
    }
    void challenge16(uint8_t out16[16]) {  // challengeBytes(16) (:215-240)
        uint8_t d[32];
        hashWith(nullptr, 0, d);
        std::memcpy(out16, d, 16);
    }
    Fr challengeScalarFull() {  // :279-312: the 16 bytes reversed, read little-endian = the digest prefix as a big-endian u128, to Montgomery
        uint8_t b[16];
        challenge16(b);
        uint64_t hi = 0, lo = 0;
        for (int i = 0; i < 8; i++) hi = (hi << 8) | b[i];
        for (int i = 8; i < 16; i++) lo = (lo << 8) | b[i];
        Fr raw{{lo, hi, 0, 0}}, r2{{Fr::R2[0], Fr::R2[1], Fr::R2[2], Fr::R2[3]}};
        return raw.mul(r2);
    }
    Fr challengeScalar() {  // :264-266,332-390: 125-bit mask, stored as RAW Montgomery limbs [0, 0, lo, hi] (MontU128Challenge)
        uint8_t b[16];
        challenge16(b);
        uint64_t hi = 0, lo = 0;  // the reversed buffer read big-endian = the digest prefix as a LITTLE-endian u128 (unlike challengeScalarFull)
        for (int i = 7; i >= 0; i--) lo = (lo << 8) | b[i];
        for (int i = 15; i >= 8; i--) hi = (hi << 8) | b[i];
        hi &= (1ULL << 61) - 1;
        return Fr{{0, 0, lo, hi}};
    }

private:
    void hashWith(const uint8_t *payload, size_t n, uint8_t *digest_out) {  // hasher() (:76-87) + payload, updateState (:90-93)
        std::vector<uint8_t> buf(64 + n, 0);
        std::memcpy(buf.data(), state, 32);
        buf[60] = (uint8_t)(n_rounds >> 24); buf[61] = (uint8_t)(n_rounds >> 16); buf[62] = (uint8_t)(n_rounds >> 8); buf[63] = (uint8_t)n_rounds;
        if (n) std::memcpy(buf.data() + 64, payload, n);
        blake2b256(buf.data(), buf.size(), state);
        n_rounds += 1;
        if (digest_out) std::memcpy(digest_out, state, 32);
    }
};

struct SumcheckVerificationFailed : std::runtime_error {
    SumcheckVerificationFailed() : std::runtime_error("SumcheckVerificationFailed") {}
};

struct Sumcheck {
    struct Round {
        UniPoly poly;
    };
    class Prover {  // src/subprotocols/mod.zig:50-134 — the polynomial lives on the GPU
    public:
        explicit Prover(const DensePolynomial &p) : round(0) {
            check(zg_sumcheck_open(reinterpret_cast<const uint64_t *>(p.evaluations.data()), p.evaluations.size(), ZG_SC_HIGH_HALF, &s_),
                  "zg_sumcheck_open");
        }
        ~Prover() { zg_sumcheck_close(s_); }
        Prover(const Prover &) = delete;
        Round nextRound() {  // :69-109 -> coefficients [g(0), g(1) - g(0)]
            Fr g0, g1;
            check(zg_sumcheck_round_sums(s_, g0.limbs, g1.limbs), "zg_sumcheck_round_sums");
            Round r;
            r.poly.coeffs = {g0, g1.sub(g0)};
            return r;
        }
        void receiveChallenge(const Fr &c) {  // :112-122
            check(zg_sumcheck_bind(s_, c.limbs), "zg_sumcheck_bind");
            round++;
        }
        bool isComplete() const { return zg_sumcheck_len(s_) == 1; }
        Fr getFinalEval() const {  // :130-133
            Fr f;
            check(zg_sumcheck_final(s_, f.limbs), "zg_sumcheck_final");
            return f;
        }
        size_t round;

    private:
        zg_sc_t s_ = nullptr;
    };
    struct Verifier {  // :137-244 (toy Fiat-Shamir mixer, host side as in the reference)
        Fr claim;
        size_t round = 0;
        std::vector<Fr> challenges;
        explicit Verifier(const Fr &c) : claim(c) {}
        Fr deriveChallenge(const Round &rd) const {  // :211-243
            uint64_t h = 0x9e3779b97f4a7c15ULL;
            h ^= (uint64_t)round;
            h *= 0xff51afd7ed558ccdULL;
            for (uint64_t limb : claim.limbs) { h ^= limb; h *= 0xc4ceb9fe1a85ec53ULL; }
            for (const Fr &c : rd.poly.coeffs)
                for (uint64_t limb : c.limbs) { h ^= limb; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; }
            h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33;
            return Fr::fromU64(h);
        }
        Fr verifyRound(const Round &rd) {  // :165-207
            Fr sum = rd.poly.evaluate(Fr::zero()).add(rd.poly.evaluate(Fr::one()));
            if (!sum.eql(claim)) throw SumcheckVerificationFailed();
            Fr ch = deriveChallenge(rd);
            challenges.push_back(ch);
            claim = rd.poly.evaluate(ch);
            round++;
            return ch;
        }
    };
    struct Proof {
        Fr claim;
        std::vector<Round> rounds;
        std::vector<Fr> final_point;
        Fr final_eval;
    };
};

struct SumcheckResult {
    Sumcheck::Proof proof;
    bool result;
};

// runSumcheck with the verifier on the host, one device round trip per round: the shape every prover with a real
// (Keccak/Blake2b) transcript has. Same outputs as runSumcheck below.
inline SumcheckResult runSumcheckInteractive(const DensePolynomial &polynomial) {  // src/subprotocols/mod.zig:302-354
    SumcheckResult out;
    Fr claim = Fr::zero();
    if (polynomial.num_vars == 0) {
        claim = polynomial.evaluations[0];
    } else {  // claim = sum of all evaluations (:306-309) = g0 + g1 of round 0
        zg_sc_t s = nullptr;
        check(zg_sumcheck_open(reinterpret_cast<const uint64_t *>(polynomial.evaluations.data()), polynomial.evaluations.size(),
                               ZG_SC_HIGH_HALF, &s), "zg_sumcheck_open");
        Fr g0, g1;
        int rc = zg_sumcheck_round_sums(s, g0.limbs, g1.limbs);
        zg_sumcheck_close(s);
        check(rc, "zg_sumcheck_round_sums");
        claim = g0.add(g1);
    }
    Sumcheck::Prover prover(polynomial);
    Sumcheck::Verifier verifier(claim);
    for (size_t i = 0; i < polynomial.num_vars; i++) {
        Sumcheck::Round rd = prover.nextRound();
        Fr ch = verifier.verifyRound(rd);
        prover.receiveChallenge(ch);
        out.proof.rounds.push_back(rd);
    }
    out.proof.claim = claim;
    out.proof.final_point = verifier.challenges;
    out.proof.final_eval = prover.getFinalEval();
    out.result = verifier.claim.eql(out.proof.final_eval);
    return out;
}

// runSumcheck (src/subprotocols/mod.zig:302-354): prover AND toy verifier on the device (zg_run_sumcheck), no PCIe
// crossing between rounds.
inline SumcheckResult runSumcheck(const DensePolynomial &polynomial) {
    SumcheckResult out;
    size_t v = polynomial.num_vars;
    std::vector<uint64_t> rounds(8 * v + 1), chal(4 * v + 1);
    uint8_t result = 0;
    int rc = zg_run_sumcheck(reinterpret_cast<const uint64_t *>(polynomial.evaluations.data()), polynomial.evaluations.size(),
                             out.proof.claim.limbs, rounds.data(), chal.data(), out.proof.final_eval.limbs, &result);
    if (rc == ZG_ERR_VERIFY) throw SumcheckVerificationFailed();
    check(rc, "zg_run_sumcheck");
    for (size_t i = 0; i < v; i++) {
        Sumcheck::Round rd;
        Fr c0, c1, ch;
        std::memcpy(c0.limbs, &rounds[8 * i], 32);
        std::memcpy(c1.limbs, &rounds[8 * i + 4], 32);
        std::memcpy(ch.limbs, &chal[4 * i], 32);
        rd.poly.coeffs = {c0, c1};
        out.proof.rounds.push_back(rd);
        out.proof.final_point.push_back(ch);
    }
    out.result = result != 0;
    return out;
}

}  // namespace zolt
