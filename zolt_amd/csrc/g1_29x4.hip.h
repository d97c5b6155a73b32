// g1_29x4.hip.h — the lazy-limb group law executed by FOUR adjacent lanes per point operation.
//
// Why. The tail of an MSM (per-bit tree sums, the 2^b doubling chains, the final tree) is a chain of ~35 dependent point
// operations. One lane needs ~3300 dependent instructions per addition, and a lone wave issues at best every other
// slot, so an addition takes 8-9 us: the tail costs ~0.4 ms whatever the MSM size, which is the whole cost of a short
// MSM. The products inside one addition are mostly independent, so a quad of lanes computes them side by side:
//   add : {U1,U2,S1,S2} -> {P^2, R^2, ZZ1*ZZ2, ZZZ1*ZZZ2} -> {PPP, Q, ZZ3} -> {Y3 (two products, one reduction), ZZZ3}
//   dbl : {U^2, X^2} -> {U*V, X*V, M^2, V*ZZ} -> {Y3, ZZZ3}
// four (three) product levels instead of fourteen (nine) products. Every lane of the quad holds the full operands and
// receives the full result (replicated), so call sites need no data layout change; the partial results travel by DPP
// quad broadcasts (v_mov_b32 quad_perm), not through LDS. Formulas, bounds and exceptional cases are those of
// g1_29.hip.h (xyzz29_add / xyzz29_dbl): the result is the same group element, and the MSM output stays bit-identical
// because it is canonicalised at the end.
#pragma once
#include "g1_29.hip.h"

namespace zg {

template <int K>
ZG_DEV u32 quad_bcast_u32(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, K | (K << 2) | (K << 4) | (K << 6), 0xf, 0xf, false);
}
template <int K>
ZG_DEV F29 quad_bcast(const F29 &v) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = quad_bcast_u32<K>(v.l[i]);
    return r;
}
// operand of lane q out of four candidates. Written with masks, not ?: — a conditional over struct members is turned into a
// choice between two ADDRESSES by the optimiser, which forces the operands (whole points) into scratch memory.
ZG_DEV F29 quad_sel(uint32_t q, const F29 &a0, const F29 &a1, const F29 &a2, const F29 &a3) {
    const u32 m0 = q == 0 ? ~0u : 0u, m1 = q == 1 ? ~0u : 0u, m2 = q == 2 ? ~0u : 0u, m3 = q == 3 ? ~0u : 0u;
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (a0.l[i] & m0) | (a1.l[i] & m1) | (a2.l[i] & m2) | (a3.l[i] & m3);
    return r;
}
ZG_DEV F29 f29_pick(bool c, const F29 &a, const F29 &b) {
    const u32 m = c ? ~0u : 0u;
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (a.l[i] & m) | (b.l[i] & ~m);
    return r;
}
ZG_DEV F29 f29_zero() {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}

// 2*P by a quad; q = lane & 3. All four lanes pass the same p and receive the same result.
ZG_DEV XYZZ29 xyzz29_dbl4(const XYZZ29 &p, uint32_t q) {
    if (xyzz29_is_identity(p)) return p;
    F29 U = f29_times2(p.y);
    // level 1: q0: V = U^2, q1: XX = X^2 (q2, q3 repeat q0's work; their result is not used)
    F29 s1 = f29_pick(q == 1, p.x, U);
    F29 t1 = f29_mul(s1, s1);
    F29 V = quad_bcast<0>(t1), XX = quad_bcast<1>(t1);
    F29 M = f29_times3(XX);  // < 4.8p
    // level 2: q0: W = U*V, q1: S = X*V, q2: MM = M^2, q3: ZZ3 = V*ZZ
    F29 t2 = f29_mul(quad_sel(q, U, p.x, M, V), quad_sel(q, V, V, M, p.zz));
    F29 W = quad_bcast<0>(t2), S = quad_bcast<1>(t2), MM = quad_bcast<2>(t2);
    XYZZ29 r;
    r.zz = quad_bcast<3>(t2);
    r.x = f29_sub4_2c(MM, S);  // < 5.6p
    // level 3: q0: Y3 = M*(S - X3) + W*(4p - Y); q1: ZZZ3 = W*ZZZ (as a two-product form with a zero second product)
    F29 z = f29_zero();
    F29 t3 = f29_mul2(f29_pick(q == 0, M, W), f29_pick(q == 0, f29_sub7(S, r.x), p.zzz), f29_pick(q == 0, W, z), f29_pick(q == 0, f29_neg4(p.y), z));
    r.y = quad_bcast<0>(t3);
    r.zzz = quad_bcast<1>(t3);
    return r;
}

// acc += (px, py) by a quad; (px, py) an affine point in lazy form, never infinity (xyzz29_madd)
ZG_DEV void xyzz29_madd4(XYZZ29 &a, bool &inf, const F29 &px, const F29 &py, uint32_t q) {
    if (inf) {
        a.x = px; a.y = py;
#pragma unroll
        for (int i = 0; i < 9; i++) { a.zz.l[i] = Fp29::ONE[i]; a.zzz.l[i] = Fp29::ONE[i]; }
        inf = false;
        return;
    }
    // level 1: q0: U2 = px*ZZ, q1: S2 = py*ZZZ (q2, q3 mirror them)
    F29 t1 = f29_mul(f29_pick((q & 1) == 0, px, py), f29_pick((q & 1) == 0, a.zz, a.zzz));
    F29 U2 = quad_bcast<0>(t1), S2 = quad_bcast<1>(t1);
    F29 Pp = f29_sub7(U2, a.x);
    F29 R = f29_sub4(S2, a.y);
    if (f29_is_zero_modp(Pp)) {  // same x: P == acc (double) or P == -acc (infinity) — rare; every lane takes the complete single-lane path
        XYZZ s = xyzz29_to_std(a, false);
        Affine pt;
        pt.x = f29_to_fp(px); pt.y = f29_to_fp(py);
        xyzz29_from_std(xyzz_madd(s, pt), a, inf);
        return;
    }
    // level 2: q0: PP = P^2, q1: RR = R^2
    F29 s2 = f29_pick((q & 1) == 0, Pp, R);
    F29 t2 = f29_mul(s2, s2);
    F29 PP = quad_bcast<0>(t2), RR = quad_bcast<1>(t2);
    // level 3: q0: PPP = P*PP, q1: Q = X1*PP, q2: ZZ3 = ZZ*PP
    F29 t3 = f29_mul(quad_sel(q, Pp, a.x, a.zz, a.zz), PP);
    F29 PPP = quad_bcast<0>(t3), Q = quad_bcast<1>(t3);
    a.zz = quad_bcast<2>(t3);
    F29 X3 = f29_x3(RR, PPP, Q);
    // level 4: q0: Y3 = R*(Q - X3) + (4p - Y1)*PPP; q1: ZZZ3 = ZZZ*PPP
    F29 z = f29_zero();
    F29 t4 = f29_mul2(f29_pick(q == 0, R, a.zzz), f29_pick(q == 0, f29_sub7(Q, X3), PPP), f29_pick(q == 0, f29_neg4(a.y), z), f29_pick(q == 0, PPP, z));
    a.y = quad_bcast<0>(t4);
    a.zzz = quad_bcast<1>(t4);
    a.x = X3;
}

// a + b by a quad (complete, as xyzz29_add)
ZG_DEV XYZZ29 xyzz29_add4(const XYZZ29 &a, const XYZZ29 &b, uint32_t q) {
    if (xyzz29_is_identity(a)) return b;
    if (xyzz29_is_identity(b)) return a;
    // level 1: U1 = X1*ZZ2, U2 = X2*ZZ1, S1 = Y1*ZZZ2, S2 = Y2*ZZZ1
    F29 t1 = f29_mul(quad_sel(q, a.x, b.x, a.y, b.y), quad_sel(q, b.zz, a.zz, b.zzz, a.zzz));
    F29 U1 = quad_bcast<0>(t1), U2 = quad_bcast<1>(t1), S1 = quad_bcast<2>(t1), S2 = quad_bcast<3>(t1);
    F29 Pp = f29_sub2(U2, U1);
    F29 R = f29_sub2(S2, S1);
    if (f29_is_zero_modp(Pp)) {  // same x: doubling or inverse points (uniform over the quad: all four hold the same values)
        if (f29_is_zero_modp(R)) return xyzz29_dbl4(a, q);
        return xyzz29_identity();
    }
    // level 2: PP = P^2, RR = R^2, ZZm = ZZ1*ZZ2, ZZZm = ZZZ1*ZZZ2
    F29 t2 = f29_mul(quad_sel(q, Pp, R, a.zz, a.zzz), quad_sel(q, Pp, R, b.zz, b.zzz));
    F29 PP = quad_bcast<0>(t2), RR = quad_bcast<1>(t2), ZZm = quad_bcast<2>(t2), ZZZm = quad_bcast<3>(t2);
    // level 3: PPP = P*PP, Q = U1*PP, ZZ3 = ZZm*PP
    F29 t3 = f29_mul(quad_sel(q, Pp, U1, ZZm, ZZm), PP);
    F29 PPP = quad_bcast<0>(t3), Q = quad_bcast<1>(t3);
    XYZZ29 r;
    r.zz = quad_bcast<2>(t3);
    r.x = f29_x3(RR, PPP, Q);
    // level 4: q0: Y3 = R*(Q - X3) + (2p - S1)*PPP; q1: ZZZ3 = ZZZm*PPP
    F29 z = f29_zero();
    F29 t4 = f29_mul2(f29_pick(q == 0, R, ZZZm), f29_pick(q == 0, f29_sub7(Q, r.x), PPP), f29_pick(q == 0, f29_neg2(S1), z), f29_pick(q == 0, PPP, z));
    r.y = quad_bcast<0>(t4);
    r.zzz = quad_bcast<1>(t4);
    return r;
}

}  // namespace zg
