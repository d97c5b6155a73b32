// g1.hip.h — BN254 G1 group law for gfx950 (device side).
//
// The reference accumulates in Jacobian coordinates (/root/reference/src/msm/mod.zig:
// 145-329: double = dbl-2009-l, addAffine = madd-2007-bl shape, add = add-2007-bl) and
// only ever exposes the final AFFINE point (toAffine :178-189). Affine coordinates of a
// group element are unique canonical field values, so any complete group law gives the
// same bytes. The device uses extended-Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; EFD shortw/xyzz, a = 0): a mixed add is 8M+2S
// (vs 7M+4S) and needs no squaring trick, which suits a mul-only ALU.
//
// Completeness (the part a GPU shortcut must not drop, SURVEY §7 "hard parts"): every
// add handles acc = inf, P = acc (-> double), P = -acc (-> inf) exactly like the
// reference's addAffine/add edge cases (msm/mod.zig:229-232,258-267,311-320).
#pragma once
#include "field.hip.h"

namespace zg {

struct Affine {  // 64 B in HBM: x limbs then y limbs (Montgomery). Infinity is carried out of band.
    Fp x, y;
};

struct XYZZ {  // 128 B; identity <=> zz == 0
    Fp x, y, zz, zzz;

    ZG_DEV static XYZZ identity() {
        XYZZ r;
        r.x = Fp::zero(); r.y = Fp::zero(); r.zz = Fp::zero(); r.zzz = Fp::zero();
        return r;
    }
    ZG_DEV bool is_identity() const { return zz.is_zero(); }
    ZG_DEV static XYZZ from_affine(const Affine &p) {
        XYZZ r;
        r.x = p.x; r.y = p.y; r.zz = Fp::one(); r.zzz = Fp::one();
        return r;
    }
};

ZG_DEV Affine affine_load(const void *p) {
    Affine a;
    a.x = fe_load<FpParams>(p);
    a.y = fe_load<FpParams>(reinterpret_cast<const char *>(p) + 32);
    return a;
}
ZG_DEV void affine_store(void *p, const Affine &a) {
    fe_store(p, a.x);
    fe_store(reinterpret_cast<char *>(p) + 32, a.y);
}
ZG_DEV XYZZ xyzz_load(const void *p) {
    const char *c = reinterpret_cast<const char *>(p);
    XYZZ r;
    r.x = fe_load<FpParams>(c); r.y = fe_load<FpParams>(c + 32);
    r.zz = fe_load<FpParams>(c + 64); r.zzz = fe_load<FpParams>(c + 96);
    return r;
}
ZG_DEV void xyzz_store(void *p, const XYZZ &v) {
    char *c = reinterpret_cast<char *>(p);
    fe_store(c, v.x); fe_store(c + 32, v.y); fe_store(c + 64, v.zz); fe_store(c + 96, v.zzz);
}

// 2*(x,y) for an affine point (mdbl-2008-s-1). y = 0 cannot occur on BN254 G1 (odd prime
// order); it would yield zz = 0 = identity, matching AffinePoint.double (msm/mod.zig:122).
ZG_DEV XYZZ xyzz_dbl_affine(const Affine &p) {
    Fp U = fe_dbl(p.y);
    Fp V = fe_sqr(U);
    Fp W = fe_mul(U, V);
    Fp S = fe_mul(p.x, V);
    Fp xx = fe_sqr(p.x);
    Fp M = fe_add(fe_dbl(xx), xx);
    XYZZ r;
    r.x = fe_sub(fe_sub(fe_sqr(M), S), S);
    r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, p.y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// 2*P (dbl-2008-s-1); identity stays identity (reference: msm/mod.zig:196)
ZG_DEV XYZZ xyzz_dbl(const XYZZ &p) {
    if (p.is_identity()) return p;
    Fp U = fe_dbl(p.y);
    Fp V = fe_sqr(U);
    Fp W = fe_mul(U, V);
    Fp S = fe_mul(p.x, V);
    Fp xx = fe_sqr(p.x);
    Fp M = fe_add(fe_dbl(xx), xx);
    XYZZ r;
    r.x = fe_sub(fe_sub(fe_sqr(M), S), S);
    r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, p.y));
    r.zz = fe_mul(V, p.zz);
    r.zzz = fe_mul(W, p.zzz);
    return r;
}

// acc + P, P affine and not infinity (madd-2008-s) — the MSM inner-loop unit
// (reference: addAffine, msm/mod.zig:229-274).
ZG_DEV XYZZ xyzz_madd(const XYZZ &a, const Affine &p) {
    if (a.is_identity()) return XYZZ::from_affine(p);
    Fp U2 = fe_mul(p.x, a.zz);
    Fp S2 = fe_mul(p.y, a.zzz);
    Fp Pp = fe_sub(U2, a.x);
    Fp R = fe_sub(S2, a.y);
    if (Pp.is_zero()) {
        if (R.is_zero()) return xyzz_dbl_affine(p);  // same point
        return XYZZ::identity();                      // opposite points
    }
    Fp PP = fe_sqr(Pp);
    Fp PPP = fe_mul(Pp, PP);
    Fp Q = fe_mul(a.x, PP);
    XYZZ r;
    r.x = fe_sub(fe_sub(fe_sub(fe_sqr(R), PPP), Q), Q);
    r.y = fe_sub(fe_mul(R, fe_sub(Q, r.x)), fe_mul(a.y, PPP));
    r.zz = fe_mul(a.zz, PP);
    r.zzz = fe_mul(a.zzz, PPP);
    return r;
}

// a + b (add-2008-s), complete (reference: add, msm/mod.zig:277-327)
ZG_DEV XYZZ xyzz_add(const XYZZ &a, const XYZZ &b) {
    if (a.is_identity()) return b;
    if (b.is_identity()) return a;
    Fp U1 = fe_mul(a.x, b.zz);
    Fp U2 = fe_mul(b.x, a.zz);
    Fp S1 = fe_mul(a.y, b.zzz);
    Fp S2 = fe_mul(b.y, a.zzz);
    Fp Pp = fe_sub(U2, U1);
    Fp R = fe_sub(S2, S1);
    if (Pp.is_zero()) {
        if (R.is_zero()) return xyzz_dbl(a);
        return XYZZ::identity();
    }
    Fp PP = fe_sqr(Pp);
    Fp PPP = fe_mul(Pp, PP);
    Fp Q = fe_mul(U1, PP);
    XYZZ r;
    r.x = fe_sub(fe_sub(fe_sub(fe_sqr(R), PPP), Q), Q);
    r.y = fe_sub(fe_mul(R, fe_sub(Q, r.x)), fe_mul(S1, PPP));
    r.zz = fe_mul(fe_mul(a.zz, b.zz), PP);
    r.zzz = fe_mul(fe_mul(a.zzz, b.zzz), PPP);
    return r;
}

ZG_DEV XYZZ xyzz_neg(const XYZZ &a) {
    XYZZ r = a;
    r.y = fe_neg(a.y);
    return r;
}

// XYZZ -> affine (reference: toAffine, msm/mod.zig:178-189; identity -> {0,0,inf}).
// 1/Z = ZZ/ZZZ, x = X/Z^2, y = Y/ZZZ.
ZG_DEV bool xyzz_to_affine(const XYZZ &p, Affine &out) {
    if (p.is_identity()) {
        out.x = Fp::zero(); out.y = Fp::zero();
        return true;  // infinity
    }
    Fp izzz = fe_inv_safegcd(p.zzz);
    Fp iz = fe_mul(izzz, p.zz);
    Fp izz = fe_sqr(iz);
    out.x = fe_mul(p.x, izz);
    out.y = fe_mul(p.y, izzz);
    return false;
}

// XYZZ -> the reference's Jacobian record (X, Y, Z) with the same affine image; identity
// is written as (1,1,0) like ProjectivePoint.identity (msm/mod.zig:154-160). Used for the
// per-GPU partial that crosses the RCCL all-gather.
//   Z := ZZZ/ZZ·ZZ^2... we simply take Z = ZZ, X' = X*ZZ, Y' = Y*ZZZ:  X'/Z^2 = X/ZZ, Y'/Z^3 = Y*ZZZ/ZZ^3 = Y/ZZZ.
ZG_DEV void xyzz_to_jacobian(const XYZZ &p, Fp &X, Fp &Y, Fp &Z) {
    if (p.is_identity()) {
        X = Fp::one(); Y = Fp::one(); Z = Fp::zero();
        return;
    }
    X = fe_mul(p.x, p.zz);
    Y = fe_mul(p.y, p.zzz);
    Z = p.zz;
}
// Jacobian (X,Y,Z) -> XYZZ: ZZ = Z^2, ZZZ = Z^3
ZG_DEV XYZZ xyzz_from_jacobian(const Fp &X, const Fp &Y, const Fp &Z) {
    XYZZ r;
    r.x = X; r.y = Y;
    r.zz = fe_sqr(Z);
    r.zzz = fe_mul(r.zz, Z);
    return r;
}

}  // namespace zg
