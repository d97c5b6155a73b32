// field.hip.h — BN254 Fr / Fp Montgomery arithmetic for gfx950 (device side).
//
// Representation at the ABI and in HBM: 4 x u64 little-endian limbs, Montgomery
// form, R = 2^256, canonical (< modulus) — the reference's in-memory format
// (/root/reference/src/field/mod.zig:131,583-584). On the device the same 32
// bytes are viewed as 8 x u32 limbs: CDNA4 has no 64x64->128 multiplier, the
// widest integer multiply-add is v_mad_u64_u32 (32x32+64).
//
// Every function returns canonical values, so results are bit-identical to the
// reference's montgomeryMul / add / sub (field/mod.zig:269-308, 402-435,
// 735-816) — the value a*b*R^-1 mod m in [0, m) is unique whatever limb width
// computes it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>


namespace zg {

typedef uint32_t u32;
typedef uint64_t u64;

#define ZG_DEV __device__ __forceinline__

// ---- field parameters (u32 limbs, little-endian); values from field/mod.zig:16-41,51-75
struct FrParams {
    static constexpr u32 MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr u32 ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                   0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr u32 R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                  0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr u32 R3[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu,
                                  0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu};  // 2^768 mod MOD
    static constexpr u32 INV = 0xefffffffu;  // -MOD^-1 mod 2^32 (low word of BN254_INV)
    static constexpr u32 MINV30 = 0x10000001u;  // MOD^-1 mod 2^30 (fe_inv_safegcd)
};
struct FpParams {
    static constexpr u32 MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr u32 ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                   0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr u32 R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                  0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    static constexpr u32 R3[8] = {0xda1530dfu, 0xb1cd6dafu, 0xa7283db6u, 0x62f210e6u,
                                  0x0ada0afbu, 0xef7f0b0cu, 0x2d592544u, 0x20fd6e90u};  // 2^768 mod MOD
    static constexpr u32 INV = 0xe4866389u;  // low word of BN254_FP_INV
    static constexpr u32 MINV30 = 0x1b799c77u;  // MOD^-1 mod 2^30 (fe_inv_safegcd)
};

template <class P>
struct Fe {
    u32 l[8];

    ZG_DEV static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = 0;
        return r;
    }
    ZG_DEV static Fe one() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = P::ONE[i];
        return r;
    }
    ZG_DEV bool is_zero() const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i];
        return o == 0;
    }
    ZG_DEV bool eq(const Fe &b) const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
};

// 16-byte vector load/store of one element (two global_load_dwordx4)
template <class P>
ZG_DEV Fe<P> fe_load(const void *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fe<P> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
template <class P>
ZG_DEV void fe_store(void *p, const Fe<P> &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// The carry chains below are written with __builtin_addc / __builtin_subc: they compile to one v_addc_co_u32 / v_subb_co_u32 per limb.
// The round-3 form ((u64)a + b + carry, carry = t >> 32) was compiled as 64-bit arithmetic — a v_sub_co, a v_subb_co of the high half,
// a sign extension, a v_lshl_add_u64 and a register copy per limb: ~80 instructions for fe_sub instead of 24 (sc_fold's inner loop,
// tools/exp, round 4).
// r = a - MOD if a >= MOD else a   (a < 2*MOD)
template <class P>
ZG_DEV Fe<P> fe_reduce_once(const Fe<P> &a) {
    Fe<P> d;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.l[i] = __builtin_subc(a.l[i], P::MOD[i], borrow, &borrow);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
    return r;
}

// field/mod.zig:402-417 / :782-798. MOD < 2^254 so a+b never carries out of 256 bits.
template <class P>
ZG_DEV Fe<P> fe_add(const Fe<P> &a, const Fe<P> &b) {
    Fe<P> s;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s.l[i] = __builtin_addc(a.l[i], b.l[i], carry, &carry);
    return fe_reduce_once(s);
}

// field/mod.zig:420-435 / :801-816
template <class P>
ZG_DEV Fe<P> fe_sub(const Fe<P> &a, const Fe<P> &b) {
    Fe<P> d;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.l[i] = __builtin_subc(a.l[i], b.l[i], borrow, &borrow);
    const u32 mask = 0u - borrow;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.l[i] = __builtin_addc(d.l[i], P::MOD[i] & mask, carry, &carry);
    return d;
}

// field/mod.zig:494-497 / :944-947 (neg(0) = 0)
template <class P>
ZG_DEV Fe<P> fe_neg(const Fe<P> &a) {
    return fe_sub(Fe<P>::zero(), a);
}

template <class P>
ZG_DEV Fe<P> fe_dbl(const Fe<P> &a) {
    return fe_add(a, a);
}

// Montgomery product a*b*R^-1 mod MOD (field/mod.zig:269-308 ≡ :735-779), CIOS over
// 8 x 32-bit words. MOD < 2^254, inputs < MOD  =>  every intermediate t < 2*MOD < 2^255,
// so the 9th word never carries further and one conditional subtract finishes.
template <class P>
ZG_DEV Fe<P> fe_mul(const Fe<P> &a, const Fe<P> &b) {
    u32 t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u64 s = (u64)a.l[i] * b.l[j] + t[j] + c;
            t[j] = (u32)s;
            c = s >> 32;
        }
        u32 t8 = t[8] + (u32)c;
        u32 m = t[0] * P::INV;
        u64 s = (u64)m * P::MOD[0] + t[0];
        c = s >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            s = (u64)m * P::MOD[j] + t[j] + c;
            t[j - 1] = (u32)s;
            c = s >> 32;
        }
        s = (u64)t8 + c;
        t[7] = (u32)s;
        t[8] = (u32)(s >> 32);
    }
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return fe_reduce_once(r);
}

template <class P>
ZG_DEV Fe<P> fe_sqr(const Fe<P> &a) {
    return fe_mul(a, a);
}

// a*R^-1: Montgomery -> canonical integer (field/mod.zig:187-189 / :642-645)
template <class P>
ZG_DEV Fe<P> fe_from_mont(const Fe<P> &a) {
    Fe<P> one = Fe<P>::zero();
    one.l[0] = 1;
    return fe_mul(a, one);
}
// raw 256-bit integer (may be >= MOD) -> Montgomery (field/mod.zig:171-184: one CIOS by R^2;
// a < 2^256, R2 < MOD  =>  t < 2*MOD still holds)
template <class P>
ZG_DEV Fe<P> fe_to_mont(const Fe<P> &a) {
    Fe<P> r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.l[i] = P::R2[i];
    return fe_mul(a, r2);
}

// Fermat inverse a^(MOD-2), LSB-first like field/mod.zig:500-518 / :955-983.
// inverse(0) returns 0 (the reference returns null; callers test is_zero first).
template <class P>
ZG_DEV Fe<P> fe_inv(const Fe<P> &a) {
    Fe<P> result = Fe<P>::one(), base = a;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        u32 e = P::MOD[w] - (w == 0 ? 2u : 0u);  // MOD[0] >= 2, no borrow
#pragma unroll 1
        for (int bit = 0; bit < 32; bit++) {
            if ((e >> bit) & 1u) result = fe_mul(result, base);
            base = fe_sqr(base);
        }
    }
    return a.is_zero() ? Fe<P>::zero() : result;
}

// ---- fast inversion: binary extended Euclid on the raw limbs, then one Montgomery fix-up.
// The reference inverts by Fermat (381 multiplications, fe_inv above). A GPU lane runs one
// multiplication in ~1 us when it is alone on its SIMD, so the inversion that ends every MSM
// (toAffine) would cost ~0.4 ms; the shift/subtract Euclid below needs ~1/6 of the instructions.
// Same value: for Montgomery input aR it returns a^-1 R = (aR)^-1 * R^2 = montmul((aR)^-1, R^3).
template <class P>
ZG_DEV bool limbs_ge(const u32 *a, const u32 *b) {  // a >= b
    bool ge = true;
#pragma unroll
    for (int i = 0; i < 8; i++) ge = (a[i] > b[i]) || (a[i] == b[i] && ge);
    return ge;
}
ZG_DEV void limbs_sub(u32 *a, const u32 *b) {  // a -= b (no borrow out expected)
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 s = (u64)a[i] - b[i] - borrow;
        a[i] = (u32)s;
        borrow = (u32)(s >> 32) & 1u;
    }
}
// x = x/2 mod MOD for x in [0, MOD): if odd add MOD first (x + MOD < 2^255 fits)
template <class P>
ZG_DEV void limbs_half_mod(u32 *x) {
    u32 mask = 0u - (x[0] & 1u);
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 t = (u64)x[i] + (P::MOD[i] & mask) + carry;
        x[i] = (u32)t;
        carry = (u32)(t >> 32);
    }
#pragma unroll
    for (int i = 0; i < 7; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
    x[7] = (x[7] >> 1) | (carry << 31);
}
ZG_DEV void limbs_shr1(u32 *x) {
#pragma unroll
    for (int i = 0; i < 7; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
    x[7] >>= 1;
}
ZG_DEV bool limbs_is_one(const u32 *x) {
    u32 o = x[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < 8; i++) o |= x[i];
    return o == 0;
}

template <class P>
ZG_DEV Fe<P> fe_inv_fast(const Fe<P> &a) {
    if (a.is_zero()) return Fe<P>::zero();
    u32 u[8], v[8];
    Fe<P> x1 = Fe<P>::zero(), x2 = Fe<P>::zero();
    x1.l[0] = 1;
#pragma unroll
    for (int i = 0; i < 8; i++) { u[i] = a.l[i]; v[i] = P::MOD[i]; }
    // invariants: x1*a == u, x2*a == v (mod MOD); gcd(a, MOD) = 1
    while (!limbs_is_one(u) && !limbs_is_one(v)) {
        while (!(u[0] & 1u)) { limbs_shr1(u); limbs_half_mod<P>(x1.l); }
        while (!(v[0] & 1u)) { limbs_shr1(v); limbs_half_mod<P>(x2.l); }
        if (limbs_ge<P>(u, v)) { limbs_sub(u, v); x1 = fe_sub(x1, x2); }
        else { limbs_sub(v, u); x2 = fe_sub(x2, x1); }
    }
    Fe<P> x = limbs_is_one(u) ? x1 : x2;
    Fe<P> r3;
#pragma unroll
    for (int i = 0; i < 8; i++) r3.l[i] = P::R3[i];
    return fe_mul(x, r3);
}

// ---- Inversion by Bernstein-Yang division steps ("safegcd"), in the batched variable-time form popularised by libsecp256k1's
// modinv32: the state (f, g) = (MOD, x) and the Bezout coefficients (d, e) are 9 signed 30-bit limbs; 30 division steps at a
// time are run on the low 32 bits of f and g only and collected in a 2x2 matrix of 31-bit entries, which is then applied to
// the four big numbers with 32x32->64 multiply-adds. ~20 batches of ~300 instructions replace the ~500 carry-chained iterations of
// a binary Euclid: on one GPU lane (where a dependent instruction costs 4-8 cycles) that is ~4x faster, and
// the single inversion at the end of an MSM was the longest item of a short MSM's tail. Value: x^-1 for the integer x;
// Montgomery form is restored with one product by R^3.
struct S30 {
    int32_t v[9];
};
template <class P>
ZG_DEV S30 s30_modulus() {
    S30 m;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int o = 30 * i, wi = o / 32, sh = o % 32;
        u32 x = P::MOD[wi] >> sh;
        if (sh > 2 && wi + 1 < 8) x |= P::MOD[wi + 1] << (32 - sh);
        m.v[i] = (int32_t)(i < 8 ? (x & 0x3fffffffu) : x);
    }
    return m;
}
ZG_DEV S30 s30_from_words(const u32 *w) {
    S30 m;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int o = 30 * i, wi = o / 32, sh = o % 32;
        u32 x = w[wi] >> sh;
        if (sh > 2 && wi + 1 < 8) x |= w[wi + 1] << (32 - sh);
        m.v[i] = (int32_t)(i < 8 ? (x & 0x3fffffffu) : x);
    }
    return m;
}
// limbs in [0, 2^30), value < 2^256
ZG_DEV void s30_to_words(const S30 &a, u32 *w) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int o = 32 * j, li = o / 30, sh = o % 30;
        u32 x = (u32)a.v[li] >> sh;
        if (li + 1 < 9) x |= (u32)a.v[li + 1] << (30 - sh);
        if (sh > 28 && li + 2 < 9) x |= (u32)a.v[li + 2] << (60 - sh);
        w[j] = x;
    }
}

template <class P>
ZG_DEV Fe<P> fe_inv_safegcd(const Fe<P> &a) {
    if (a.is_zero()) return Fe<P>::zero();
    const int32_t M30 = 0x3fffffff;
    const S30 mod = s30_modulus<P>();
    S30 f = mod, g = s30_from_words(a.l), d, e;
#pragma unroll
    for (int i = 0; i < 9; i++) { d.v[i] = 0; e.v[i] = 0; }
    e.v[0] = 1;
    int32_t eta = -1;
    for (int batch = 0; batch < 40; batch++) {  // 590 steps bound the loop for 256-bit inputs; it ends when g == 0
        // ---- 30 division steps on the low words -> transition matrix [u v; q r]
        u32 u = 1, v = 0, q = 0, r = 1;
        u32 fl = (u32)f.v[0] | ((u32)f.v[1] << 30), gl = (u32)g.v[0] | ((u32)g.v[1] << 30);
        int i = 30;
        for (;;) {
            int zeros = __builtin_ctz(gl | (0xffffffffu << i));  // trailing zeros of g, at most i
            gl >>= zeros;
            u <<= zeros;
            v <<= zeros;
            eta -= zeros;
            i -= zeros;
            if (i == 0) break;
            if (eta < 0) {  // swap f and g (with a sign), and the matrix rows with them
                u32 t;
                eta = -eta;
                t = fl; fl = gl; gl = 0u - t;
                t = u; u = q; q = 0u - t;
                t = v; v = r; r = 0u - t;
            }
            // eta >= 0: cancel min(eta + 1, i, 6) low bits of g with a multiple of f; -g/f mod 2^6 = g*f*(f^2 - 2)
            int limit = (eta + 1) > i ? i : (eta + 1);
            u32 m = (0xffffffffu >> (32 - limit)) & 63u;
            u32 w = (fl * gl * (fl * fl - 2u)) & m;
            gl += fl * w;
            q += u * w;
            r += v * w;
        }
        const int32_t tu = (int32_t)u, tv = (int32_t)v, tq = (int32_t)q, tr = (int32_t)r;
        // ---- (d, e) <- t * (d, e) / 2^30 mod MOD: multiples md, me of MOD make the low 30 bits vanish
        {
            int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
            int32_t md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
            int64_t cd = (int64_t)tu * d.v[0] + (int64_t)tv * e.v[0];
            int64_t ce = (int64_t)tq * d.v[0] + (int64_t)tr * e.v[0];
            md -= (int32_t)((P::MINV30 * (u32)cd + (u32)md) & (u32)M30);
            me -= (int32_t)((P::MINV30 * (u32)ce + (u32)me) & (u32)M30);
            cd += (int64_t)mod.v[0] * md;
            ce += (int64_t)mod.v[0] * me;
            cd >>= 30;
            ce >>= 30;
            int32_t d0 = d.v[0], e0 = e.v[0];
            (void)d0; (void)e0;
#pragma unroll
            for (int k = 1; k < 9; k++) {
                int32_t di = d.v[k], ei = e.v[k];
                cd += (int64_t)tu * di + (int64_t)tv * ei + (int64_t)mod.v[k] * md;
                ce += (int64_t)tq * di + (int64_t)tr * ei + (int64_t)mod.v[k] * me;
                d.v[k - 1] = (int32_t)cd & M30;
                cd >>= 30;
                e.v[k - 1] = (int32_t)ce & M30;
                ce >>= 30;
            }
            d.v[8] = (int32_t)cd;
            e.v[8] = (int32_t)ce;
        }
        // ---- (f, g) <- t * (f, g) / 2^30 (exact)
        u32 gnz = 0;
        {
            int64_t cf = (int64_t)tu * f.v[0] + (int64_t)tv * g.v[0];
            int64_t cg = (int64_t)tq * f.v[0] + (int64_t)tr * g.v[0];
            cf >>= 30;
            cg >>= 30;
#pragma unroll
            for (int k = 1; k < 9; k++) {
                int32_t fi = f.v[k], gi = g.v[k];
                cf += (int64_t)tu * fi + (int64_t)tv * gi;
                cg += (int64_t)tq * fi + (int64_t)tr * gi;
                f.v[k - 1] = (int32_t)cf & M30;
                cf >>= 30;
                g.v[k - 1] = (int32_t)cg & M30;
                cg >>= 30;
                gnz |= (u32)g.v[k - 1];
            }
            f.v[8] = (int32_t)cf;
            g.v[8] = (int32_t)cg;
            gnz |= (u32)g.v[8];
        }
        if (gnz == 0) break;
    }
    // f = +-1 and d * x = f (mod MOD): bring d from (-2 MOD, MOD) to [0, MOD), negated when f = -1
    {
        int32_t cond_add = d.v[8] >> 31, cond_neg = f.v[8] >> 31;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            d.v[k] += mod.v[k] & cond_add;
            d.v[k] = (d.v[k] ^ cond_neg) - cond_neg;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            d.v[k + 1] += d.v[k] >> 30;
            d.v[k] &= M30;
        }
        cond_add = d.v[8] >> 31;
#pragma unroll
        for (int k = 0; k < 9; k++) d.v[k] += mod.v[k] & cond_add;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            d.v[k + 1] += d.v[k] >> 30;
            d.v[k] &= M30;
        }
    }
    Fe<P> x, r3;
    s30_to_words(d, x.l);
#pragma unroll
    for (int i = 0; i < 8; i++) r3.l[i] = P::R3[i];
    return fe_mul(x, r3);  // (aR)^-1 * R^3 * R^-1 = a^-1 R
}

typedef Fe<FrParams> Fr;
typedef Fe<FpParams> Fp;

}  // namespace zg
