// g1_29.hip.h — the MSM inner loop's group law on lazy 29-bit-limb field elements (fp29.hip.h).
//
// Same madd-2008-s formulas as g1.hip.h:xyzz_madd (the reference's addAffine, src/msm/mod.zig:229-274),
// with the reduction bounds tracked statically instead of conditional subtractions:
//
//   class            bound      where
//   point x, y       < 1.1 p    table rows (f29_from_fp outputs); negated y = 2p - y <= 2p
//   M (mul output)   < 1.6 p    every f29_mul / f29_sqr below has input classes A*B <= 101
//   acc.x            < 6.6 p    X3 = R^2 + 5p - PPP - 2Q
//   acc.y            < 3.6 p    Y3 = M + 2p - M
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
    if (f29_is_zero_modp(Pp)) {  // same x: P == acc (double) or P == -acc (infinity) — rare, take the complete path
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
    F29 Y3 = f29_sub2(f29_mul(R, f29_sub7(Q, X3)), f29_mul(a.y, PPP));
    a.zz = f29_mul(a.zz, PP);
    a.zzz = f29_mul(a.zzz, PPP);
    a.x = X3;
    a.y = Y3;
}

}  // namespace zg
