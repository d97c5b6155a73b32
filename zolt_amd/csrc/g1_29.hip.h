// g1_29.hip.h — the MSM inner loop's group law on lazy 29-bit-limb field elements (fp29.hip.h).
//
// Same madd-2008-s formulas as g1.hip.h:xyzz_madd (the reference's addAffine, src/msm/mod.zig:229-274),
// with the reduction bounds tracked statically instead of conditional subtractions:
//
//   class            bound      where
//   point x, y       < 1.1 p    table rows (f29_from_fp outputs); negated y = 2p - y <= 2p
//   M (mul output)   < 1.6 p    every f29_mul / f29_sqr below has input classes A*B <= 101
//   acc.x            < 6.6 p    X3 = R^2 + 5p - PPP - 2Q
//   acc.y            < 1.4 p    Y3 = (R*(Q - X3) + (4p - Y1)*PPP) * 2^-261, one reduction: (5.6*8.6 + 4*1.6)/168.9 + 1
//   P  = U2 + 7p - X1  < 8.6 p,  R = S2 + 4p - Y1 < 5.6 p,  Q + 7p - X3 < 8.6 p
//
// The exceptional cases of a mixed add (acc == P -> double, acc == -P -> infinity) are detected
// exactly (f29_is_zero_modp) and handed to the canonical-form code path, which is complete.
#pragma once
#include "fp29.hip.h"
#include "g1.hip.h"

namespace zg {

struct XYZZ29 {
    F29 x, y, zz, zzz;
};

ZG_DEV XYZZ xyzz29_to_std(const XYZZ29 &a, bool inf) {
    if (inf) return XYZZ::identity();
    XYZZ r;
    r.x = f29_to_fp(a.x); r.y = f29_to_fp(a.y); r.zz = f29_to_fp(a.zz); r.zzz = f29_to_fp(a.zzz);
    return r;
}
ZG_DEV void xyzz29_from_std(const XYZZ &s, XYZZ29 &a, bool &inf) {
    inf = s.is_identity();
    a.x = f29_from_fp(s.x); a.y = f29_from_fp(s.y); a.zz = f29_from_fp(s.zz); a.zzz = f29_from_fp(s.zzz);
}

// acc += (px, py); (px, py) is an affine point in lazy form, never infinity
ZG_DEV void xyzz29_madd(XYZZ29 &a, bool &inf, const F29 &px, const F29 &py) {
    if (inf) {
        a.x = px; a.y = py;
#pragma unroll
        for (int i = 0; i < 9; i++) { a.zz.l[i] = Fp29::ONE[i]; a.zzz.l[i] = Fp29::ONE[i]; }
        inf = false;
        return;
    }
    F29 U2 = f29_mul(px, a.zz);
    F29 S2 = f29_mul(py, a.zzz);
    F29 Pp = f29_sub7(U2, a.x);
    F29 R = f29_sub4(S2, a.y);
#ifdef ZG_EXP_NOSLOW
    if (false) {
#else
    if (f29_is_zero_modp(Pp)) {  // same x: P == acc (double) or P == -acc (infinity) — rare, take the complete path
#endif
        XYZZ s = xyzz29_to_std(a, false);
        Affine q;
        q.x = f29_to_fp(px); q.y = f29_to_fp(py);
        xyzz29_from_std(xyzz_madd(s, q), a, inf);
        return;
    }
    F29 PP = f29_sqr(Pp);
    F29 PPP = f29_mul(Pp, PP);
    F29 Q = f29_mul(a.x, PP);
    F29 X3 = f29_x3(f29_sqr(R), PPP, Q);
    F29 Y3 = f29_mul2(R, f29_sub7(Q, X3), f29_neg4(a.y), PPP);  // R*(Q - X3) - Y1*PPP, one reduction
    a.zz = f29_mul(a.zz, PP);
    a.zzz = f29_mul(a.zzz, PPP);
    a.x = X3;
    a.y = Y3;
}

// ---- full group law on lazy elements, for the bucket-reduction kernels.
// Identity is encoded as zz = all-zero limbs (a non-identity ZZ is a non-zero field element, whose lazy
// representative cannot be 0). Classes as above: X < 6.6p, Y < 1.4p, ZZ, ZZZ < 1.6p.
ZG_DEV XYZZ29 xyzz29_identity() {
    XYZZ29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) { r.x.l[i] = 0; r.y.l[i] = 0; r.zz.l[i] = 0; r.zzz.l[i] = 0; }
    return r;
}
ZG_DEV bool xyzz29_is_identity(const XYZZ29 &a) { return f29_all_zero(a.zz); }

ZG_DEV XYZZ29 xyzz29_from_std_val(const XYZZ &s) {
    if (s.is_identity()) return xyzz29_identity();
    XYZZ29 a;
    a.x = f29_from_fp(s.x); a.y = f29_from_fp(s.y); a.zz = f29_from_fp(s.zz); a.zzz = f29_from_fp(s.zzz);
    return a;
}
ZG_DEV XYZZ xyzz29_to_std_val(const XYZZ29 &a) { return xyzz29_to_std(a, xyzz29_is_identity(a)); }

// 2*P (dbl-2008-s-1)
ZG_DEV XYZZ29 xyzz29_dbl(const XYZZ29 &p) {
    if (xyzz29_is_identity(p)) return p;
    F29 U = f29_times2(p.y);            // < 7.2p
    F29 V = f29_sqr(U);
    F29 W = f29_mul(U, V);
    F29 S = f29_mul(p.x, V);
    F29 M = f29_times3(f29_sqr(p.x));   // < 4.8p
    XYZZ29 r;
    r.x = f29_sub4_2c(f29_sqr(M), S);   // < 5.6p
    r.y = f29_mul2(M, f29_sub7(S, r.x), W, f29_neg4(p.y));
    r.zz = f29_mul(V, p.zz);
    r.zzz = f29_mul(W, p.zzz);
    return r;
}

// Jacobian doubling on y^2 = x^3 + b for the table build's chains of doublings (msm_precompute_kernel): three squarings, two products and
// one two-product sum — 945 multiply-adds against the 1269 of xyzz29_dbl, which carries ZZ and ZZZ separately.
//   A = X^2, B = Y^2, D = 4 X B, E = 3 A, F = E^2:  X3 = F - 2 D,  Y3 = E (D - X3) - 8 B^2,  Z3 = 2 Y Z
// Classes (closed under the map, and an affine table row x, y < 1.1p with Z = one starts inside them): X < 5.3p, Y < 1.3p, Z < 1.1p;
// A < 1.2p, B < 1.1p, 4B < 4.1p, D < 1.2p, E < 3.6p, F < 1.1p, D + 7p - X3 < 8.2p, 8B < 8.2p:
// Y3 < ((3.6 * 8.2 + 2 * 8.2) / 168.9 + 1) p. Never called on the identity (the group has odd prime order: Y is never 0).
struct Jac29 {
    F29 x, y, z;
};
ZG_DEV Jac29 jac29_dbl(const Jac29 &p) {
    F29 A = f29_sqr(p.x);
    F29 B = f29_sqr(p.y);
    F29 B4 = f29_times4(B);
    F29 D = f29_mul(p.x, B4);
    F29 E = f29_times3(A);
    F29 F = f29_sqr(E);
    Jac29 r;
    r.x = f29_sub4_2c(F, D);  // F + 4p - 2D
    r.y = f29_mul2(E, f29_sub7(D, r.x), f29_neg2(B), f29_times2(B4));  // E (D - X3) + (2p - B) * 8B, one reduction
    r.z = f29_mul(f29_times2(p.y), p.z);
    return r;
}

// a + b (add-2008-s), complete
ZG_DEV XYZZ29 xyzz29_add(const XYZZ29 &a, const XYZZ29 &b) {
    if (xyzz29_is_identity(a)) return b;
    if (xyzz29_is_identity(b)) return a;
    F29 U1 = f29_mul(a.x, b.zz);
    F29 U2 = f29_mul(b.x, a.zz);
    F29 S1 = f29_mul(a.y, b.zzz);
    F29 S2 = f29_mul(b.y, a.zzz);
    F29 Pp = f29_sub2(U2, U1);
    F29 R = f29_sub2(S2, S1);
    if (f29_is_zero_modp(Pp)) {
        if (f29_is_zero_modp(R)) return xyzz29_dbl(a);
        return xyzz29_identity();
    }
    F29 PP = f29_sqr(Pp);
    F29 PPP = f29_mul(Pp, PP);
    F29 Q = f29_mul(U1, PP);
    XYZZ29 r;
    r.x = f29_x3(f29_sqr(R), PPP, Q);
    r.y = f29_mul2(R, f29_sub7(Q, r.x), f29_neg2(S1), PPP);
    r.zz = f29_mul(f29_mul(a.zz, b.zz), PP);
    r.zzz = f29_mul(f29_mul(a.zzz, b.zzz), PPP);
    return r;
}

// 144-byte records (4 x 9 u32), 16-byte aligned
ZG_DEV XYZZ29 xyzz29_load(const void *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint4 v = q[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    XYZZ29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) { r.x.l[i] = w[i]; r.y.l[i] = w[9 + i]; r.zz.l[i] = w[18 + i]; r.zzz.l[i] = w[27 + i]; }
    return r;
}
ZG_DEV void xyzz29_store(void *p, const XYZZ29 &v) {
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) { w[i] = v.x.l[i]; w[9 + i] = v.y.l[i]; w[18 + i] = v.zz.l[i]; w[27 + i] = v.zzz.l[i]; }
    uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
ZG_DEV XYZZ29 xyzz29_shfl_down(const XYZZ29 &v, int delta) {
    XYZZ29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.x.l[i] = __shfl_down(v.x.l[i], delta, 64);
        r.y.l[i] = __shfl_down(v.y.l[i], delta, 64);
        r.zz.l[i] = __shfl_down(v.zz.l[i], delta, 64);
        r.zzz.l[i] = __shfl_down(v.zzz.l[i], delta, 64);
    }
    return r;
}

}  // namespace zg
