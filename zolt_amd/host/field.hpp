// field.hpp — host Fr / Fp scalars (the unchanged `field` module's job above the FFI seam).
// Part of zolt_host.hpp (the C++ host mirror over include/zolt_gpu.h); included by it, after the parts it depends on.
#pragma once
#ifndef ZOLT_HOST_UMBRELLA
#error "include zolt_host.hpp"
#endif
namespace zolt {

// ---------------------------------------------------------------- host Fr (scalar use only)
struct Fr {
    uint64_t limbs[4];

    static constexpr uint64_t MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t R[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
    static constexpr uint64_t INV = 0xc2e1f593efffffffULL;

    static Fr zero() { return Fr{{0, 0, 0, 0}}; }
    static Fr one() { return Fr{{R[0], R[1], R[2], R[3]}}; }
    bool isZero() const { return (limbs[0] | limbs[1] | limbs[2] | limbs[3]) == 0; }
    bool eql(const Fr &o) const { return std::memcmp(limbs, o.limbs, 32) == 0; }

    static bool geMod(const uint64_t *a) {
        for (int i = 3; i >= 0; i--) {
            if (a[i] < MOD[i]) return false;
            if (a[i] > MOD[i]) return true;
        }
        return true;
    }
    static void subMod(uint64_t *a) {
        unsigned __int128 borrow = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 d = (unsigned __int128)a[i] - MOD[i] - borrow;
            a[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
    }
    Fr mul(const Fr &o) const {  // src/field/mod.zig:735-779
        uint64_t t[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            uint64_t carry = 0;
            for (int j = 0; j < 4; j++) {
                unsigned __int128 s = (unsigned __int128)t[j] + (unsigned __int128)limbs[i] * o.limbs[j] + carry;
                t[j] = (uint64_t)s;
                carry = (uint64_t)(s >> 64);
            }
            t[4] += carry;
            uint64_t m = t[0] * INV;
            unsigned __int128 s0 = (unsigned __int128)t[0] + (unsigned __int128)m * MOD[0];
            carry = (uint64_t)(s0 >> 64);
            for (int j = 1; j < 4; j++) {
                unsigned __int128 s = (unsigned __int128)t[j] + (unsigned __int128)m * MOD[j] + carry;
                t[j - 1] = (uint64_t)s;
                carry = (uint64_t)(s >> 64);
            }
            unsigned __int128 fs = (unsigned __int128)t[4] + carry;
            t[3] = (uint64_t)fs;
            t[4] = (uint64_t)(fs >> 64);
        }
        Fr r{{t[0], t[1], t[2], t[3]}};
        if (t[4] != 0 || geMod(r.limbs)) subMod(r.limbs);
        return r;
    }
    Fr add(const Fr &o) const {  // :782-798
        Fr r;
        unsigned __int128 carry = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 s = (unsigned __int128)limbs[i] + o.limbs[i] + carry;
            r.limbs[i] = (uint64_t)s;
            carry = s >> 64;
        }
        if (carry || geMod(r.limbs)) subMod(r.limbs);
        return r;
    }
    Fr sub(const Fr &o) const {  // :801-816
        Fr r;
        unsigned __int128 borrow = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 d = (unsigned __int128)limbs[i] - o.limbs[i] - borrow;
            r.limbs[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
        if (borrow) {
            unsigned __int128 carry = 0;
            for (int i = 0; i < 4; i++) {
                unsigned __int128 s = (unsigned __int128)r.limbs[i] + MOD[i] + carry;
                r.limbs[i] = (uint64_t)s;
                carry = s >> 64;
            }
        }
        return r;
    }
    bool inverse(Fr &out) const {  // :955-983 — Fermat, a^(p-2); false for zero (Zig: null)
        if (isZero()) return false;
        uint64_t e[4] = {MOD[0] - 2, MOD[1], MOD[2], MOD[3]};
        Fr result = one(), base = *this;
        for (int i = 0; i < 256; i++) {
            if ((e[i / 64] >> (i % 64)) & 1) result = result.mul(base);
            base = base.mul(base);
        }
        out = result;
        return true;
    }
    static Fr fromU64(uint64_t n) {  // :617-622
        Fr a{{n, 0, 0, 0}}, r2{{R2[0], R2[1], R2[2], R2[3]}};
        return a.mul(r2);
    }
    static Fr fromBytes(const uint8_t *bytes) {  // :625-639: 32 little-endian bytes (may exceed the modulus), times R^2
        Fr a, r2{{R2[0], R2[1], R2[2], R2[3]}};
        for (int i = 0; i < 4; i++) {
            uint64_t v = 0;
            for (int b = 7; b >= 0; b--) v = (v << 8) | bytes[8 * i + b];
            a.limbs[i] = v;
        }
        return a.mul(r2);
    }
};

// Fp values cross the host only as opaque limbs (coordinates of points)
struct Fp {
    uint64_t limbs[4];
    static constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t TWO[4] = {0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL, 0x1c14ef83340fbe5eULL};
};

}  // namespace zolt
