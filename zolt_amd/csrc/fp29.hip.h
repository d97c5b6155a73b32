// fp29.hip.h — BN254 Fp in 9 x 29-bit redundant limbs, for the MSM bucket-accumulation inner loop.
//
// Why a second representation. field.hip.h multiplies 8 x 32-bit limbs; every 32x32 product then
// needs carry handling, and on gfx950 v_add_co/v_addc_co and 64-bit adds cost as much as a
// v_mad_u64_u32 (4 cycles per wave-instruction, tools/microbench.hip). With 29-bit limbs a column of
// the schoolbook product holds at most 9 + 9 products of < 2^58, which fits a 64-bit accumulator, so
// the whole multiplication is a stream of v_mad_u64_u32 with NO carry instructions until one
// shift-and-add per column at the end: ~225 VALU instructions instead of ~560.
//
// Values are in Montgomery form with R' = 2^261 and are kept LAZILY reduced: a value is any
// representative in [0, k*p) for a small k tracked per formula (see xyzz29_madd); p < 2^254 leaves
// 7 spare bits, so a Montgomery product of inputs < A*p and < B*p is < (A*B/168.9 + 1)*p with no final
// subtraction. Limbs are "near-normalised" (< 2^29 + 8) except the top limb. Nothing in this format
// ever leaves the MSM: bucket sums are converted back to canonical 8 x 32-bit Montgomery-2^256
// (field.hip.h) before they are stored, so results stay bit-identical to the reference.
#pragma once
#include "field.hip.h"

namespace zg {

struct F29 {
    u32 l[9];
};

struct Fp29 {
    static constexpr u32 MASK = 0x1fffffffu;
    static constexpr u32 P[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr u32 NINV = 0x04866389u;   // -p^-1 mod 2^29
    static constexpr u32 PINV0 = 0x1b799c77u;  //  p^-1 mod 2^29
    static constexpr u32 ONE[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                                   0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};  // 2^261 mod p
    // ext (x*2^256) -> int (x*2^261): mont29(ext, K_IN), K_IN = 2^266 mod p
    static constexpr u32 K_IN[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u,
                                    0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
    // int -> ext: mont29(int, K_OUT), K_OUT = 2^256 mod p
    static constexpr u32 K_OUT[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u,
                                     0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
    // k*p with limbs 0..7 raised by 2^30 (2^31 for 5p) and the excess borrowed from the next limb, so
    // that (bias - b) never underflows limb-wise for near-normalised b (3 of them for 5p)
    static constexpr u32 BIAS2P[9] = {0x50f9fa8eu, 0x4208c16bu, 0x58e5469cu, 0x45aa459fu, 0x4b0bb2eeu,
                                      0x45b6817fu, 0x414dc280u, 0x5cb84c66u, 0x0060c89au};
    static constexpr u32 BIAS4P[9] = {0x41f3f51cu, 0x441182d9u, 0x51ca8d3au, 0x4b548b41u, 0x561765deu,
                                      0x4b6d0300u, 0x429b8502u, 0x597098ceu, 0x00c19137u};
    static constexpr u32 BIAS5P[9] = {0x9a70f263u, 0x8515e38du, 0x8e3d3087u, 0x8e29ae10u, 0x8b9d3f54u,
                                      0x8e4843bfu, 0x83426641u, 0x87ccbf00u, 0x00f1f584u};
    static constexpr u32 BIAS7P[9] = {0x4b6aecf1u, 0x471ea4fdu, 0x47227727u, 0x53d3f3b4u, 0x56a8f246u,
                                      0x53fec542u, 0x449028c5u, 0x44850b6au, 0x0152be23u};
};

// one parallel carry step: limbs < 2^32 in, limbs < 2^29 + 8 out (top limb keeps the rest)
ZG_DEV F29 f29_carry(const F29 &x) {
    F29 r;
    r.l[0] = x.l[0] & Fp29::MASK;
#pragma unroll
    for (int i = 1; i < 8; i++) r.l[i] = (x.l[i] & Fp29::MASK) + (x.l[i - 1] >> 29);
    r.l[8] = x.l[8] + (x.l[7] >> 29);
    return r;
}

// 29-bit-limb constants of the scalar field, for the poly kernels' mixed-format multiply (fr_mul29)
struct Fr29 {
    static constexpr u32 MASK = 0x1fffffffu;
    static constexpr u32 P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr u32 NINV = 0x0fffffffu;  // -r^-1 mod 2^29
    // fr29_prescale(R^2 mod r): fr_mul29(x, R2PRE) = x * R^2 * 2^-256 = toMontgomery(x)
    static constexpr u32 R2PRE[9] = {0x142db4dfu, 0x19d6990eu, 0x1472f48cu, 0x06dbe7e3u, 0x0b84d579u,
                                     0x10f9faf7u, 0x121f4380u, 0x17a112deu, 0x001275c7u};
};

// Montgomery product a*b*2^-261 mod p (lazy): limbs of a, b < 2^30; output limbs exactly < 2^29,
// value < (A*B/168.9 + 1)*p for a < A*p, b < B*p.
// Column-serial (Comba) order: column k collects its product terms and the reduction terms of the earlier
// quotient digits in ONE 64-bit accumulator whose initial value is the carry out of column k-1, so the
// carry propagation costs a shift only (no 64-bit add): every v_mad_u64_u32 takes its addend for free.
// A column holds <= 9 + 9 products of < 2^58 plus a carry < 2^36 (27 for the two-product form): < 2^63.
// c + a*b as one v_mad_u64_u32. ZG_F29_ASM pins the instruction (and with it the accumulation order) with inline
// assembly; the default leaves instruction selection to the compiler.
#ifdef ZG_F29_ASM
ZG_DEV u64 mad64(u32 a, u32 b, u64 c) {
    u64 d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
}
ZG_DEV u64 mad64k(u32 a, u32 k, u64 c) {  // k: a compile-time constant, kept in an SGPR
    u64 d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c) : "vcc");
    return d;
}
#else
ZG_DEV u64 mad64(u32 a, u32 b, u64 c) { return c + (u64)a * b; }
ZG_DEV u64 mad64k(u32 a, u32 k, u64 c) { return c + (u64)a * k; }
#endif

#ifndef ZG_F29_ROWWISE
template <class C29, int NPROD>
ZG_DEV F29 f29t_mulsum(const F29 &a, const F29 &b, const F29 &c, const F29 &d) {
    u32 m[9];
    F29 r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) {
            acc = mad64(a.l[i], b.l[k - i], acc);
            if (NPROD == 2) acc = mad64(c.l[i], d.l[k - i], acc);
        }
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 9 ? k - 1 : 8); i++) acc = mad64k(m[i], C29::P[k - i], acc);
        if (k < 9) {
            m[k] = ((u32)acc * C29::NINV) & C29::MASK;
            acc = mad64k(m[k], C29::P[0], acc);
        } else {
            r.l[k - 9] = (u32)acc & C29::MASK;
        }
        acc >>= 29;
    }
    r.l[8] = (u32)acc;
    return r;
}
template <class C29>
ZG_DEV F29 f29t_mul(const F29 &a, const F29 &b) { return f29t_mulsum<C29, 1>(a, b, a, b); }

ZG_DEV F29 f29_mul(const F29 &a, const F29 &b) { return f29t_mul<Fp29>(a, b); }
// (a*b + c*d) * 2^-261 with ONE reduction: value < ((A*B + C*D)/168.9 + 1)*p
ZG_DEV F29 f29_mul2(const F29 &a, const F29 &b, const F29 &c, const F29 &d) { return f29t_mulsum<Fp29, 2>(a, b, c, d); }

ZG_DEV F29 f29_sqr(const F29 &a) {
    u32 m[9], d2[9];
    F29 r;
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) d2[i] = a.l[i] << 1;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); 2 * i < k; i++) acc = mad64(d2[i], a.l[k - i], acc);
        if ((k & 1) == 0) acc = mad64(a.l[k / 2], a.l[k / 2], acc);
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 9 ? k - 1 : 8); i++) acc = mad64k(m[i], Fp29::P[k - i], acc);
        if (k < 9) {
            m[k] = ((u32)acc * Fp29::NINV) & Fp29::MASK;
            acc = mad64k(m[k], Fp29::P[0], acc);
        } else {
            r.l[k - 9] = (u32)acc & Fp29::MASK;
        }
        acc >>= 29;
    }
    r.l[8] = (u32)acc;
    return r;
}
#else
template <class C29>
ZG_DEV F29 f29t_mul(const F29 &a, const F29 &b) {
    u64 c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (u64)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 m = ((u32)c[i] * C29::NINV) & C29::MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (u64)m * C29::P[j];
        c[i + 1] += c[i] >> 29;
    }
    F29 r;
#pragma unroll
    for (int k = 9; k < 17; k++) {
        r.l[k - 9] = (u32)c[k] & C29::MASK;
        c[k + 1] += c[k] >> 29;
    }
    r.l[8] = (u32)c[17];
    return r;
}

ZG_DEV F29 f29_mul(const F29 &a, const F29 &b) { return f29t_mul<Fp29>(a, b); }

ZG_DEV F29 f29_sqr(const F29 &a) {
    u64 c[18];
    u32 d[9];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c[2 * i] += (u64)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < 9; j++) c[i + j] += (u64)d[i] * a.l[j];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 m = ((u32)c[i] * Fp29::NINV) & Fp29::MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (u64)m * Fp29::P[j];
        c[i + 1] += c[i] >> 29;
    }
    F29 r;
#pragma unroll
    for (int k = 9; k < 17; k++) {
        r.l[k - 9] = (u32)c[k] & Fp29::MASK;
        c[k + 1] += c[k] >> 29;
    }
    r.l[8] = (u32)c[17];
    return r;
}

ZG_DEV F29 f29_mul2(const F29 &a, const F29 &b, const F29 &c, const F29 &d) {
    u64 t[18];
#pragma unroll
    for (int k = 0; k < 18; k++) t[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) t[i + j] += (u64)a.l[i] * b.l[j] + (u64)c.l[i] * d.l[j];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 m = ((u32)t[i] * Fp29::NINV) & Fp29::MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) t[i + j] += (u64)m * Fp29::P[j];
        t[i + 1] += t[i] >> 29;
    }
    F29 r;
#pragma unroll
    for (int k = 9; k < 17; k++) {
        r.l[k - 9] = (u32)t[k] & Fp29::MASK;
        t[k + 1] += t[k] >> 29;
    }
    r.l[8] = (u32)t[17];
    return r;
}
#endif

// a + K*p - b, near-normalised; needs b < K*p (with margin) and near-normalised limbs
#define ZG_F29_SUB(NAME, BIAS)                                        \
    ZG_DEV F29 NAME(const F29 &a, const F29 &b) {                     \
        F29 t;                                                        \
        _Pragma("unroll") for (int i = 0; i < 9; i++) t.l[i] = a.l[i] + Fp29::BIAS[i] - b.l[i]; \
        return f29_carry(t);                                          \
    }
ZG_F29_SUB(f29_sub2, BIAS2P)
ZG_F29_SUB(f29_sub4, BIAS4P)
ZG_F29_SUB(f29_sub7, BIAS7P)
#undef ZG_F29_SUB

// 2p - y (negation of an affine y < 2p)
ZG_DEV F29 f29_neg2(const F29 &y) {
    F29 t;
#pragma unroll
    for (int i = 0; i < 9; i++) t.l[i] = Fp29::BIAS2P[i] - y.l[i];
    return f29_carry(t);
}

// 4p - y (y < 4p, near-normalised)
ZG_DEV F29 f29_neg4(const F29 &y) {
    F29 t;
#pragma unroll
    for (int i = 0; i < 9; i++) t.l[i] = Fp29::BIAS4P[i] - y.l[i];
    return f29_carry(t);
}

// a + 5p - b - 2c  (b, c exactly normalised mul outputs)
ZG_DEV F29 f29_x3(const F29 &a, const F29 &b, const F29 &c) {
    F29 t;
#pragma unroll
    for (int i = 0; i < 9; i++) t.l[i] = a.l[i] + Fp29::BIAS5P[i] - b.l[i] - 2u * c.l[i];
    return f29_carry(t);
}

// a + 4p - 2c (c an exactly normalised mul output < 1.6p)
ZG_DEV F29 f29_sub4_2c(const F29 &a, const F29 &c) {
    F29 t;
#pragma unroll
    for (int i = 0; i < 9; i++) t.l[i] = a.l[i] + Fp29::BIAS4P[i] - 2u * c.l[i];
    return f29_carry(t);
}
// small multiples, carried: inputs near-normalised
ZG_DEV F29 f29_times2(const F29 &a) {
    F29 t;
#pragma unroll
    for (int i = 0; i < 9; i++) t.l[i] = a.l[i] << 1;
    return f29_carry(t);
}
ZG_DEV F29 f29_times4(const F29 &a) {
    F29 t;
#pragma unroll
    for (int i = 0; i < 9; i++) t.l[i] = a.l[i] << 2;
    return f29_carry(t);
}
ZG_DEV F29 f29_times3(const F29 &a) {
    F29 t;
#pragma unroll
    for (int i = 0; i < 9; i++) t.l[i] = a.l[i] * 3u;
    return f29_carry(t);
}
ZG_DEV bool f29_all_zero(const F29 &a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.l[i];
    return o == 0;
}

// x == 0 (mod p) for a near-normalised x < 16p whose limb 0 is exact (< 2^29: any f29_carry output).
// If x = k*p then k = x_0 * p^-1 mod 2^29; anything else passes this filter with probability 2^-25.
ZG_DEV bool f29_is_zero_modp(const F29 &x) {
    u32 k = (x.l[0] * Fp29::PINV0) & Fp29::MASK;
    if (k > 16u) return false;
    u32 carry = 0;
    u64 kp = 0;
    bool eq = true;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 v = x.l[i] + carry;
        kp += (u64)k * Fp29::P[i];
        if (i < 8) {
            eq = eq && ((v & Fp29::MASK) == ((u32)kp & Fp29::MASK));
            carry = v >> 29;
            kp >>= 29;
        } else {
            eq = eq && (v == (u32)kp);
        }
    }
    return eq;
}

// 256-bit little-endian words (value < 2^256) <-> 9 x 29-bit limbs
ZG_DEV F29 f29_unpack(const u32 *w) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int o = 29 * i, wi = o / 32, sh = o % 32;
        u32 v = w[wi] >> sh;
        if (sh > 3 && wi + 1 < 8) v |= w[wi + 1] << (32 - sh);
        r.l[i] = (i < 8) ? (v & Fp29::MASK) : v;
    }
    return r;
}
// limbs must be exactly normalised and the value < 2^256
ZG_DEV void f29_pack(const F29 &x, u32 *w) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int o = 32 * j, li = o / 29, sh = o % 29;  // word j starts at bit `sh` of limb li
        u32 v = x.l[li] >> sh;
        if (li + 1 < 9) v |= x.l[li + 1] << (29 - sh);
        if (sh > 26 && li + 2 < 9) v |= x.l[li + 2] << (58 - sh);
        w[j] = v;
    }
}

// canonical Montgomery-2^256 element (field.hip.h) -> lazy Montgomery-2^261
ZG_DEV F29 f29_from_fp(const Fp &a) {
    F29 k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = Fp29::K_IN[i];
    return f29_mul(f29_unpack(a.l), k);
}
// lazy (value < 16p) -> canonical Montgomery-2^256 element
ZG_DEV Fp f29_to_fp(const F29 &x) {
    F29 k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = Fp29::K_OUT[i];
    F29 t = f29_mul(x, k);  // < 1.1p, limbs exact
    // conditional subtract of p (29-bit borrow chain)
    F29 d;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 v = t.l[i] - Fp29::P[i] - borrow;
        if (i < 8) {
            borrow = v >> 31;
            d.l[i] = v & Fp29::MASK;
        } else {
            borrow = v >> 31;
            d.l[i] = v;
        }
    }
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = borrow ? t.l[i] : d.l[i];
    Fp out;
    f29_pack(r, out.l);
    return out;
}

// ---- mixed-format product for the poly kernels. Table entries live in HBM in the ABI format (canonical
// Montgomery-2^256). When one factor is shared by many products (a sumcheck challenge, an eq-table row factor)
// it is pre-scaled ONCE:  y' = 32*y mod r  as 29-bit limbs; then for any canonical x
//     mont261(x, y') = x*y*2^5*2^-261 = x*y*2^-256 (mod r)
// is exactly the reference's montgomeryMul(x, y) — computed with the carry-free 29-bit columns (~300
// instructions incl. unpack, one conditional subtraction and repack, instead of ~560).
ZG_DEV F29 fr29_prescale(const Fr &y) {
    // 32 * y as an unreduced lazy operand (< 32 r, limbs < 2^29): a 5-bit shift across the limbs. Round 3 reduced it to the canonical
    // 32 y mod r with five modular doublings (~225 dependent instructions in front of every launch's first product); the multiplier
    // does not need that — for a canonical x the product is < (32 / 168.9 + 1) r < 2 r either way, and the canonical result is the same.
    F29 yu = f29_unpack(y.l), ys;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 lo = i ? (yu.l[i - 1] >> 24) : 0u;
        ys.l[i] = (i < 8) ? (((yu.l[i] << 5) & Fp29::MASK) | lo) : ((yu.l[i] << 5) | lo);
    }
    // opaque from here: a prescaled factor is reused by every product of a loop, and the compiler otherwise carries the shift-and-mask
    // expressions into those products and multiplies by their pieces separately (198 multiply-adds + ~100 copies instead of 162)
#pragma unroll
    for (int i = 0; i < 9; i++) asm volatile("" : "+v"(ys.l[i]));
    return ys;
}
ZG_DEV Fr fr_mul29(const Fr &x, const F29 &y_pre) {
    F29 t = f29t_mul<Fr29>(f29_unpack(x.l), y_pre);  // < (1/168.9 + 1) r, limbs exact
    F29 d;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 v = t.l[i] - Fr29::P[i] - borrow;
        borrow = v >> 31;
        d.l[i] = (i < 8) ? (v & Fr29::MASK) : v;
    }
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = borrow ? t.l[i] : d.l[i];
    Fr out;
    f29_pack(r, out.l);
    return out;
}

// The same product for two VARIABLE canonical factors (no shared one to pre-scale): the factor 2^5 is applied to y as a
// 5-bit left shift across its 29-bit limbs — 32*y < 2^259 still fits the nine limbs, and the lazy multiplier accepts an
// operand < 32 r: the product is < (32/168.9 + 1) r < 2 r, so one conditional subtraction restores the canonical value.
// ~330 instructions against ~560 for the 32-bit-limb CIOS (fe_mul); bit-identical result.
ZG_DEV Fr fr_mul29v(const Fr &x, const Fr &y) {
    F29 yu = f29_unpack(y.l), ys;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 lo = i ? (yu.l[i - 1] >> 24) : 0u;
        ys.l[i] = (i < 8) ? (((yu.l[i] << 5) & Fr29::MASK) | lo) : ((yu.l[i] << 5) | lo);
    }
    return fr_mul29(x, ys);
}

// ---- product CHAINS of canonical scalars without leaving the 29-bit-limb form (the product-form sumcheck kernels, psc.hip).
// fr_mul29v pays an unpack of both operands, a conditional subtraction and a repack per product; in a chain v = f0 * f1 * f2 ...
// only the factors need unpacking (the later ones with the 5-bit shift that turns the 2^-261 of the lazy multiplier into the ABI's
// 2^-256), the running product stays a lazy value (< 1.3 r, exact limbs), and a SUM of chain values is kept limb-wise in 64-bit
// words — no carries, no reduction — until one multiplication by 2^261 mod r brings it back to the canonical element.
ZG_DEV F29 fr29_in(const Fr &x) { return f29_unpack(x.l); }      // first factor of a chain (value < r)
ZG_DEV F29 fr29_in_shift(const Fr &y) {                         // every later factor: 32 * y < 32 r, limbs < 2^29 (top < 2^27)
    F29 yu = f29_unpack(y.l), ys;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 lo = i ? (yu.l[i - 1] >> 24) : 0u;
        ys.l[i] = (i < 8) ? (((yu.l[i] << 5) & Fr29::MASK) | lo) : ((yu.l[i] << 5) | lo);
    }
    return ys;
}
// a < A r (A <= 4), b = fr29_in_shift(y): a * y * 2^-256 mod r as a lazy value < (32 A / 168.9 + 1) r <= 1.76 r, exact limbs
ZG_DEV F29 fr29_chain_mul(const F29 &a, const F29 &b_shifted) { return f29t_mul<Fr29>(a, b_shifted); }
// a lazy chain value (< 2 r, exact limbs) -> canonical element
ZG_DEV Fr fr29_out(const F29 &t) {
    F29 d;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 v = t.l[i] - Fr29::P[i] - borrow;
        borrow = v >> 31;
        d.l[i] = (i < 8) ? (v & Fr29::MASK) : v;
    }
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = borrow ? t.l[i] : d.l[i];
    Fr out;
    f29_pack(r, out.l);
    return out;
}
// ---- narrow shared factors. The reference's sumcheck challenges are MontU128Challenge values: the stored Montgomery element is
// [0, 0, lo, hi] (src/transcripts: challengeScalar; tests/golden/stage2_batched_rounds.json), i.e. y = H * 2^128 with H < 2^126.
// Then x * y * 2^-256 = x * H * 2^-128 = x * (H * 2^17) * 2^-145: a 9 x 5-limb product and FIVE reduction steps instead of
// 9 x 9 and nine — 90 multiply-adds instead of 162 (the reference takes the same shortcut in its own multiplier). The output is
// < x * H / 2^128 + r < 2 r with exact limbs, as fr29_out expects. FrMul picks the form once per launch from the factor itself;
// any other factor takes the full-width path, so the result never depends on the choice.
template <class C29, int NB>
ZG_DEV F29 f29t_mul_short(const F29 &a, const F29 &b) {  // b: NB limbs used
    u32 m[NB];
    F29 r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9 + NB - 1; k++) {
#pragma unroll
        for (int i = (k > NB - 1 ? k - (NB - 1) : 0); i <= (k < 8 ? k : 8); i++) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < NB ? k - 1 : NB - 1); i++) acc = mad64k(m[i], C29::P[k - i], acc);
        if (k < NB) {
            m[k] = ((u32)acc * C29::NINV) & C29::MASK;
            acc = mad64k(m[k], C29::P[0], acc);
        } else {
            r.l[k - NB] = (u32)acc & C29::MASK;
        }
        acc >>= 29;
    }
    r.l[8] = (u32)acc;
    return r;
}
struct FrMul {
    F29 p;
    bool narrow;
};
ZG_DEV FrMul frmul_prepare(const Fr &y) {
    FrMul m;
    m.narrow = (y.l[0] | y.l[1] | y.l[2] | y.l[3]) == 0;
    if (m.narrow) {  // H * 2^17 < 2^145: five limbs
        u32 w[8] = {y.l[4] << 17, (y.l[5] << 17) | (y.l[4] >> 15), (y.l[6] << 17) | (y.l[5] >> 15), (y.l[7] << 17) | (y.l[6] >> 15),
                    y.l[7] >> 15, 0u, 0u, 0u};
        m.p = f29_unpack(w);
    } else {
        m.p = fr29_prescale(y);
    }
    // opaque from here: otherwise the compiler carries the shift-and-mask expressions of the prescale into every product of the
    // launch's loop and multiplies by their pieces separately (198 multiply-adds and ~100 register copies per product instead of 162)
#pragma unroll
    for (int i = 0; i < 9; i++) asm volatile("" : "+v"(m.p.l[i]));
    return m;
}
ZG_DEV Fr frmul_apply(const Fr &x, const FrMul &m) {
    F29 xu = f29_unpack(x.l);
    F29 t;
    if (m.narrow) t = f29t_mul_short<Fr29, 5>(xu, m.p);
    else t = f29t_mul<Fr29>(xu, m.p);
    return fr29_out(t);
}

// F.fromU64(u) = u * R mod r (src/field/mod.zig:157-169) by a 9 x 3-limb product with three reduction steps:
// (2^343 mod r) * u * 2^-87 = u * 2^256, < 2 r with exact limbs. 54 multiply-adds instead of the 162 of a product by R^2.
ZG_DEV Fr fr_from_u64_29(u64 u) {
    constexpr u32 K343[9] = {0x02a7c5aeu, 0x0efc10bfu, 0x01057868u, 0x0a968059u, 0x10b7c2c3u, 0x04d6a71fu, 0x075f0711u, 0x1678993cu, 0x0024e88bu};
    F29 k, b;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        k.l[i] = K343[i];
        b.l[i] = 0;
    }
    b.l[0] = (u32)u & Fr29::MASK;
    b.l[1] = (u32)(u >> 29) & Fr29::MASK;
    b.l[2] = (u32)(u >> 58);
    return fr29_out(f29t_mul_short<Fr29, 3>(k, b));
}

// limb-wise sum of lazy chain values in 64-bit words: at most FR29_ACC_MAX values (< 2 r each) between two reductions
struct Acc29 {
    u64 l[9];
};
constexpr unsigned FR29_ACC_MAX = 64;  // 64 * 2 r < 2^261: the sum still fits the nine limbs the multiplier accepts
ZG_DEV Acc29 acc29_zero() {
    Acc29 a;
#pragma unroll
    for (int i = 0; i < 9; i++) a.l[i] = 0;
    return a;
}
ZG_DEV void acc29_add(Acc29 &a, const F29 &v) {
#pragma unroll
    for (int i = 0; i < 9; i++) a.l[i] += v.l[i];
}
// the sum as a canonical element: carry into 29-bit limbs (top limb keeps the rest, < 2^30), one lazy product by 2^261 mod r
// (X * 2^261 * 2^-261 = X mod r, < (128 / 168.9 + 1) r < 2 r), one conditional subtraction
ZG_DEV Fr acc29_reduce(const Acc29 &a) {
    constexpr u32 C261[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    F29 x, c;
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u64 v = a.l[i] + carry;
        x.l[i] = (i < 8) ? ((u32)v & Fr29::MASK) : (u32)v;
        carry = v >> 29;
        c.l[i] = C261[i];
    }
    return fr29_out(f29t_mul<Fr29>(x, c));
}

// The same sum kept in 32-bit limbs (36 registers less for four running sums): limb-wise additions with a carry pass (f29_carry)
// at least every FOUR additions — 2^29 + 8 + 4 * 2^29 < 2^32 — and the reduction below after at most FR29_ACC_MAX values.
// x: near-normalised limbs (any f29_carry output), value < 128 r.
ZG_DEV Fr fr29_sum_reduce(const F29 &x) {
    constexpr u32 C261[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    F29 c;
#pragma unroll
    for (int i = 0; i < 9; i++) c.l[i] = C261[i];
    return fr29_out(f29t_mul<Fr29>(x, c));
}

// Montgomery -> canonical integer of a scalar (fromMontgomery, src/field/mod.zig:642-645) = montgomeryMul(x, 1): the
// prescaled 1 is the constant 32, so the product half of the multiplication folds to nine shifts and only the reduction
// remains (~260 instructions instead of ~500 for the 32-bit-limb CIOS by one). Canonical output.
ZG_DEV Fr fr_from_mont29(const Fr &x) {
    F29 one_pre;
#pragma unroll
    for (int i = 0; i < 9; i++) one_pre.l[i] = i == 0 ? 32u : 0u;
    return fr_mul29(x, one_pre);
}

// table row (64 B): x, y as packed Montgomery-2^261 values (< 2^256, not necessarily < p)
ZG_DEV void f29_store_packed(void *p, const F29 &x) {
    Fp t;
    f29_pack(x, t.l);
    fe_store(p, t);
}

}  // namespace zg
