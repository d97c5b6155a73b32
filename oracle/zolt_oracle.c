/*
 * zolt_oracle.c — CPU restatement of the Zolt hot path (TEST INFRASTRUCTURE).
 *
 * This file is the parity ORACLE for the gfx950 backend. It restates, in plain
 * C with `unsigned __int128`, the algorithms of the reference's Zig CPU path
 * (MatteoMer/zolt, mounted at /root/reference) function by function; every
 * function cites the reference file:line it follows.
 *
 *   ONLY tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 *   load this library. The product (libzolt_gpu.so, zolt_amd/) never links,
 *   imports or calls it.
 *
 * Pinning: the reference is Zig-only and cannot be compiled here (no zig
 * toolchain). This restatement is pinned against the reference's own golden
 * data instead — see tests/test_oracle_golden.py:
 *   - the 64-byte bytecode commitment inside logs/zolt_proof_regular.bin
 *     (a real MSM(Fr,Fp).compute output of the reference), regenerated from
 *     examples/fibonacci.elf;
 *   - the KATs embedded in the reference's inline tests (field, msm, poly,
 *     sumcheck), and an independent Python big-int model (oracle/pymodel.py).
 *
 * Element format everywhere: 4 x u64 little-endian limbs, Montgomery form with
 * R = 2^256, canonical (< modulus) — src/field/mod.zig:131,583-584.
 * Points at this ABI: xy = 8 x u64 (x limbs then y limbs) + separate u8
 * infinity flag (the Zig struct layout is not ABI-stable, src/msm/mod.zig:19-21).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;

typedef struct {
    uint64_t mod[4], r[4], r2[4], inv;
} fparams;

/* src/field/mod.zig:16-41 (Fr) */
static const fparams FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
    {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
    0xc2e1f593efffffffULL};
/* src/field/mod.zig:51-75 (Fp) */
static const fparams FP = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
    {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
    0x87d20782e4866389ULL};

/* ------------------------------------------------------------------ field */

/* lessThanModulus — src/field/mod.zig:520-528 / :1005-1013 */
static inline int f_lt_mod(const fparams *P, const fe *a) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] < P->mod[i]) return 1;
        if (a->l[i] > P->mod[i]) return 0;
    }
    return 0;
}
/* subtractModulus — :530-541 / :1015-1026 */
static inline void f_sub_mod(const fparams *P, fe *a) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - P->mod[i] - borrow;
        a->l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}
/* addModulus — :543-554 / :1028-1039 */
static inline void f_add_mod(const fparams *P, fe *a) {
    uint64_t carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a->l[i] + P->mod[i] + carry;
        a->l[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
}

/* montgomeryMul (CIOS) — src/field/mod.zig:269-308 (generic) ≡ :735-779 (Fr) */
static inline fe f_mul(const fparams *P, const fe *a, const fe *b) {
    uint64_t t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 s = (u128)t[j] + (u128)a->l[i] * b->l[j] + carry;
            t[j] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        t[4] = t[4] + carry; /* :280-281 truncating add */
        uint64_t m = t[0] * P->inv;
        u128 s0 = (u128)t[0] + (u128)m * P->mod[0];
        carry = (uint64_t)(s0 >> 64);
        for (int j = 1; j < 4; j++) {
            u128 s = (u128)t[j] + (u128)m * P->mod[j] + carry;
            t[j - 1] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        u128 fs = (u128)t[4] + carry;
        t[3] = (uint64_t)fs;
        t[4] = (uint64_t)(fs >> 64);
    }
    fe r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] != 0 || !f_lt_mod(P, &r)) f_sub_mod(P, &r);
    return r;
}
/* add — :402-417 / :782-798 */
static inline fe f_add(const fparams *P, const fe *a, const fe *b) {
    fe r; uint64_t carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a->l[i] + b->l[i] + carry;
        r.l[i] = (uint64_t)s; carry = (uint64_t)(s >> 64);
    }
    if (carry != 0 || !f_lt_mod(P, &r)) f_sub_mod(P, &r);
    return r;
}
/* sub — :420-435 / :801-816 */
static inline fe f_sub(const fparams *P, const fe *a, const fe *b) {
    fe r; uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        r.l[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) f_add_mod(P, &r);
    return r;
}
static inline int f_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int f_eq(const fe *a, const fe *b) {
    return a->l[0] == b->l[0] && a->l[1] == b->l[1] && a->l[2] == b->l[2] && a->l[3] == b->l[3];
}
/* neg — :494-497 / :944-947 (neg(0) = 0) */
static inline fe f_neg(const fparams *P, const fe *a) {
    if (f_is_zero(a)) return *a;
    fe m = {{P->mod[0], P->mod[1], P->mod[2], P->mod[3]}};
    return f_sub(P, &m, a);
}
/* square — Fp: :443-445 (= mul(self,self)). Fr: :866-941 is a hand-rolled SOS
 * squaring whose `+%=` at :886/:908 drops a carry with probability ~2^-64; the
 * mathematically intended value a^2·R^-1 is restated here. */
static inline fe f_sqr(const fparams *P, const fe *a) { return f_mul(P, a, a); }
static inline fe f_one(const fparams *P) { fe r = {{P->r[0], P->r[1], P->r[2], P->r[3]}}; return r; }
static inline fe f_zero(void) { fe r = {{0, 0, 0, 0}}; return r; }
/* fromU64 — :164-168 / :617-622 */
static inline fe f_from_u64(const fparams *P, uint64_t n) {
    fe a = {{n, 0, 0, 0}}, r2 = {{P->r2[0], P->r2[1], P->r2[2], P->r2[3]}};
    return f_mul(P, &a, &r2);
}
/* fromMontgomery — :187-189 / :642-645 */
static inline fe f_from_mont(const fparams *P, const fe *a) {
    fe one = {{1, 0, 0, 0}};
    return f_mul(P, a, &one);
}
/* toMontgomery / fromBytes (LE; input may be >= modulus) — :171-184,192-200 / :625-652 */
static inline fe f_to_mont(const fparams *P, const fe *a) {
    fe r2 = {{P->r2[0], P->r2[1], P->r2[2], P->r2[3]}};
    return f_mul(P, a, &r2);
}
/* inverse (Fermat, LSB-first square-and-multiply) — :500-518 / :955-983.
 * returns 0 for input 0 (Zig: null). */
static int f_inv(const fparams *P, const fe *a, fe *out) {
    if (f_is_zero(a)) return 0;
    uint64_t e[4] = {P->mod[0] - 2, P->mod[1], P->mod[2], P->mod[3]};
    fe result = f_one(P), base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) result = f_mul(P, &result, &base);
        base = f_sqr(P, &base);
    }
    *out = result;
    return 1;
}

/* ---------------------------------------------------------- exported field */
#define EXPORT __attribute__((visibility("default")))
static const fparams *sel(int which) { return which ? &FP : &FR; } /* 0 = Fr, 1 = Fp */

EXPORT void zo_f_mul(int f, const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) { fe r = f_mul(sel(f), (const fe *)(a + 4 * i), (const fe *)(b + 4 * i)); memcpy(o + 4 * i, &r, 32); }
}
EXPORT void zo_f_add(int f, const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) { fe r = f_add(sel(f), (const fe *)(a + 4 * i), (const fe *)(b + 4 * i)); memcpy(o + 4 * i, &r, 32); }
}
EXPORT void zo_f_sub(int f, const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) { fe r = f_sub(sel(f), (const fe *)(a + 4 * i), (const fe *)(b + 4 * i)); memcpy(o + 4 * i, &r, 32); }
}
EXPORT void zo_f_neg(int f, const uint64_t *a, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) { fe r = f_neg(sel(f), (const fe *)(a + 4 * i)); memcpy(o + 4 * i, &r, 32); }
}
EXPORT void zo_f_sqr(int f, const uint64_t *a, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) { fe r = f_sqr(sel(f), (const fe *)(a + 4 * i)); memcpy(o + 4 * i, &r, 32); }
}
/* returns number of zero inputs (their outputs are set to 0) */
EXPORT size_t zo_f_inv(int f, const uint64_t *a, uint64_t *o, size_t n) {
    size_t z = 0;
    for (size_t i = 0; i < n; i++) {
        fe r = f_zero();
        if (!f_inv(sel(f), (const fe *)(a + 4 * i), &r)) z++;
        memcpy(o + 4 * i, &r, 32);
    }
    return z;
}
EXPORT void zo_f_from_u64(int f, const uint64_t *v, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) { fe r = f_from_u64(sel(f), v[i]); memcpy(o + 4 * i, &r, 32); }
}
EXPORT void zo_f_from_mont(int f, const uint64_t *a, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) { fe r = f_from_mont(sel(f), (const fe *)(a + 4 * i)); memcpy(o + 4 * i, &r, 32); }
}
/* raw 256-bit LE integers (possibly >= modulus) -> Montgomery; = fromBytes :171-184 */
EXPORT void zo_f_to_mont(int f, const uint64_t *a, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) { fe r = f_to_mont(sel(f), (const fe *)(a + 4 * i)); memcpy(o + 4 * i, &r, 32); }
}
/* toBytesBE — :213-237 / :665-681 */
EXPORT void zo_f_to_bytes_be(int f, const uint64_t *a, uint8_t *out32) {
    fe s = f_from_mont(sel(f), (const fe *)a);
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out32[31 - (i * 8 + b)] = (uint8_t)(s.l[i] >> (8 * b));
}

/* --------------------------------------------------------------------- G1 */
typedef struct { fe x, y; int inf; } aff;
typedef struct { fe x, y, z; } jac;

/* AffinePoint.identity — src/msm/mod.zig:24-30 */
static aff a_identity(void) { aff p; p.x = f_zero(); p.y = f_zero(); p.inf = 1; return p; }
/* ProjectivePoint.identity = (1,1,0) — :154-160 */
static jac j_identity(void) { jac p; p.x = f_one(&FP); p.y = f_one(&FP); p.z = f_zero(); return p; }
/* fromAffine — :163-170 */
static jac j_from_affine(const aff *p) {
    if (p->inf) return j_identity();
    jac r; r.x = p->x; r.y = p->y; r.z = f_one(&FP); return r;
}
/* toAffine — :178-189 */
static aff j_to_affine(const jac *p) {
    if (f_is_zero(&p->z)) return a_identity();
    fe zi;
    if (!f_inv(&FP, &p->z, &zi)) return a_identity();
    fe zi2 = f_sqr(&FP, &zi), zi3 = f_mul(&FP, &zi2, &zi);
    aff r; r.inf = 0;
    r.x = f_mul(&FP, &p->x, &zi2);
    r.y = f_mul(&FP, &p->y, &zi3);
    return r;
}
/* double (dbl-2009-l) — :195-226 */
static jac j_double(const jac *p) {
    if (f_is_zero(&p->z)) return *p;
    const fparams *F = &FP;
    fe A = f_sqr(F, &p->x), B = f_sqr(F, &p->y), C = f_sqr(F, &B);
    fe xpb = f_add(F, &p->x, &B);
    fe t = f_sqr(F, &xpb); t = f_sub(F, &t, &A); fe halfD = f_sub(F, &t, &C);
    fe D = f_add(F, &halfD, &halfD);
    fe E = f_add(F, &A, &A); E = f_add(F, &E, &A);
    fe FF = f_sqr(F, &E);
    fe twoD = f_add(F, &D, &D);
    fe X3 = f_sub(F, &FF, &twoD);
    fe c8 = f_add(F, &C, &C); for (int i = 0; i < 6; i++) c8 = f_add(F, &c8, &C); /* 8*C, :216 */
    fe dmx = f_sub(F, &D, &X3);
    fe Y3 = f_mul(F, &E, &dmx); Y3 = f_sub(F, &Y3, &c8);
    fe yz = f_mul(F, &p->y, &p->z);
    fe Z3 = f_add(F, &yz, &yz);
    jac r = {X3, Y3, Z3};
    return r;
}
/* addAffine (madd-2007-bl shape) — :229-274; edge-case order as in the reference */
static jac j_add_affine(const jac *s, const aff *o) {
    if (o->inf) return *s;
    if (f_is_zero(&s->z)) return j_from_affine(o);
    const fparams *F = &FP;
    fe z1z1 = f_sqr(F, &s->z);
    fe U2 = f_mul(F, &o->x, &z1z1);
    fe S2 = f_mul(F, &o->y, &s->z); S2 = f_mul(F, &S2, &z1z1);
    fe H = f_sub(F, &U2, &s->x);
    fe HH = f_sqr(F, &H);
    fe I = f_add(F, &HH, &HH); I = f_add(F, &I, &HH); I = f_add(F, &I, &HH);
    fe J = f_mul(F, &H, &I);
    fe d = f_sub(F, &S2, &s->y);
    fe r = f_add(F, &d, &d);
    fe V = f_mul(F, &s->x, &I);
    fe X3 = f_sqr(F, &r); X3 = f_sub(F, &X3, &J); X3 = f_sub(F, &X3, &V); X3 = f_sub(F, &X3, &V);
    fe vmx = f_sub(F, &V, &X3);
    fe y1j = f_mul(F, &s->y, &J);
    fe Y3 = f_mul(F, &r, &vmx); Y3 = f_sub(F, &Y3, &y1j); Y3 = f_sub(F, &Y3, &y1j);
    fe zph = f_add(F, &s->z, &H);
    fe Z3 = f_sqr(F, &zph); Z3 = f_sub(F, &Z3, &z1z1); Z3 = f_sub(F, &Z3, &HH);
    if (f_is_zero(&H)) {
        if (f_is_zero(&r)) return j_double(s);
        return j_identity();
    }
    jac out = {X3, Y3, Z3};
    return out;
}
/* add (add-2007-bl) — :277-327 */
static jac j_add(const jac *s, const jac *o) {
    if (f_is_zero(&s->z)) return *o;
    if (f_is_zero(&o->z)) return *s;
    const fparams *F = &FP;
    fe z1z1 = f_sqr(F, &s->z), z2z2 = f_sqr(F, &o->z);
    fe U1 = f_mul(F, &s->x, &z2z2), U2 = f_mul(F, &o->x, &z1z1);
    fe S1 = f_mul(F, &s->y, &o->z); S1 = f_mul(F, &S1, &z2z2);
    fe S2 = f_mul(F, &o->y, &s->z); S2 = f_mul(F, &S2, &z1z1);
    fe H = f_sub(F, &U2, &U1);
    fe twoH = f_add(F, &H, &H);
    fe I = f_sqr(F, &twoH);
    fe J = f_mul(F, &H, &I);
    fe d = f_sub(F, &S2, &S1);
    fe r = f_add(F, &d, &d);
    fe V = f_mul(F, &U1, &I);
    fe X3 = f_sqr(F, &r); X3 = f_sub(F, &X3, &J); X3 = f_sub(F, &X3, &V); X3 = f_sub(F, &X3, &V);
    fe vmx = f_sub(F, &V, &X3);
    fe s1j = f_mul(F, &S1, &J);
    fe Y3 = f_mul(F, &r, &vmx); Y3 = f_sub(F, &Y3, &s1j); Y3 = f_sub(F, &Y3, &s1j);
    fe zz = f_add(F, &s->z, &o->z);
    fe Z3 = f_sqr(F, &zz); Z3 = f_sub(F, &Z3, &z1z1); Z3 = f_sub(F, &Z3, &z2z2); Z3 = f_mul(F, &Z3, &H);
    if (f_is_zero(&H)) {
        if (f_is_zero(&r)) return j_double(s);
        return j_identity();
    }
    jac out = {X3, Y3, Z3};
    return out;
}
/* AffinePoint.double — :118-138 */
static aff a_double(const aff *p) {
    if (p->inf) return *p;
    if (f_is_zero(&p->y)) return a_identity();
    const fparams *F = &FP;
    fe xs = f_sqr(F, &p->x);
    fe t3 = f_add(F, &xs, &xs); t3 = f_add(F, &t3, &xs);
    fe ty = f_add(F, &p->y, &p->y), tyi;
    if (!f_inv(F, &ty, &tyi)) return a_identity();
    fe lam = f_mul(F, &t3, &tyi);
    fe x3 = f_sqr(F, &lam); x3 = f_sub(F, &x3, &p->x); x3 = f_sub(F, &x3, &p->x);
    fe d = f_sub(F, &p->x, &x3);
    fe y3 = f_mul(F, &lam, &d); y3 = f_sub(F, &y3, &p->y);
    aff r; r.x = x3; r.y = y3; r.inf = 0; return r;
}
/* AffinePoint.add — :74-103 */
static aff a_add(const aff *s, const aff *o) {
    if (s->inf) return *o;
    if (o->inf) return *s;
    const fparams *F = &FP;
    if (f_eq(&s->x, &o->x)) {
        fe ny = f_neg(F, &o->y);
        if (f_eq(&s->y, &ny)) return a_identity();
        if (f_eq(&s->y, &o->y)) return a_double(s);
    }
    fe dy = f_sub(F, &o->y, &s->y), dx = f_sub(F, &o->x, &s->x), dxi;
    if (!f_inv(F, &dx, &dxi)) return a_identity();
    fe lam = f_mul(F, &dy, &dxi);
    fe x3 = f_sqr(F, &lam); x3 = f_sub(F, &x3, &s->x); x3 = f_sub(F, &x3, &o->x);
    fe d = f_sub(F, &s->x, &x3);
    fe y3 = f_mul(F, &lam, &d); y3 = f_sub(F, &y3, &s->y);
    aff r; r.x = x3; r.y = y3; r.inf = 0; return r;
}
/* isOnCurve y^2 = x^3 + 3 — :106-115 */
static int a_on_curve(const aff *p) {
    if (p->inf) return 1;
    const fparams *F = &FP;
    fe y2 = f_sqr(F, &p->y), x3 = f_sqr(F, &p->x); x3 = f_mul(F, &x3, &p->x);
    fe b = f_from_u64(F, 3), rhs = f_add(F, &x3, &b);
    return f_eq(&y2, &rhs);
}

static aff load_aff(const uint64_t *xy, const uint8_t *inf, size_t i) {
    aff p;
    memcpy(&p.x, xy + 8 * i, 32); memcpy(&p.y, xy + 8 * i + 4, 32);
    p.inf = inf ? (inf[i] != 0) : 0;
    return p;
}
static void store_aff(const aff *p, uint64_t *xy, uint8_t *inf) {
    memcpy(xy, &p->x, 32); memcpy(xy + 4, &p->y, 32);
    if (inf) *inf = (uint8_t)p->inf;
}

/* --------------------------------------------------------------------- MSM */

/* optimalWindowSize — src/msm/mod.zig:475-484 */
EXPORT size_t zo_optimal_window_size(size_t n) {
    if (n < 8) return 1;
    if (n < 32) return 2;
    if (n < 128) return 3;
    if (n < 512) return 4;
    if (n < 2048) return 5;
    if (n < 8192) return 6;
    if (n < 32768) return 7;
    return 8;
}
/* getWindow — :441-471 (fromMontgomery per call kept, as in the reference) */
static size_t get_window(const fe *scalar, size_t window_idx, size_t c) {
    fe ns = f_from_mont(&FR, scalar);
    size_t bit_offset = window_idx * c;
    size_t limb_idx = bit_offset / 64, bit_in_limb = bit_offset % 64;
    if (limb_idx >= 4) return 0;
    uint64_t mask = (1ULL << c) - 1;
    uint64_t value = (ns.l[limb_idx] >> bit_in_limb) & mask;
    if (bit_in_limb + c > 64 && limb_idx + 1 < 4) {
        size_t remaining = bit_in_limb + c - 64;
        if (remaining > 0 && remaining <= 63 && bit_in_limb > 0) {
            uint64_t next_mask = (1ULL << remaining) - 1;
            value |= (ns.l[limb_idx + 1] & next_mask) << (64 - bit_in_limb);
        }
    }
    return (size_t)(value & mask);
}
EXPORT size_t zo_get_window(const uint64_t *scalar_mont, size_t window_idx, size_t c) {
    return get_window((const fe *)scalar_mont, window_idx, c);
}
/* scalarMul (MSB-first double-and-add) — :503-540 */
static jac scalar_mul(const aff *base, const fe *scalar) {
    if (base->inf) return j_identity();
    if (f_is_zero(scalar)) return j_identity();
    jac result = j_identity();
    fe ns = f_from_mont(&FR, scalar);
    for (int li = 3; li >= 0; li--) {
        uint64_t limb = ns.l[li];
        for (int b = 63; b >= 0; b--) {
            if (!f_is_zero(&result.z)) result = j_double(&result);
            if ((limb >> b) & 1) {
                if (f_is_zero(&result.z)) result = j_from_affine(base);
                else result = j_add_affine(&result, base);
            }
        }
    }
    return result;
}
/* naiveMSM — :487-499 */
static aff naive_msm(const uint64_t *xy, const uint8_t *inf, const uint64_t *sc, size_t n) {
    jac result = j_identity();
    for (size_t i = 0; i < n; i++) {
        aff b = load_aff(xy, inf, i);
        jac term = scalar_mul(&b, (const fe *)(sc + 4 * i));
        result = j_add(&result, &term);
    }
    return j_to_affine(&result);
}
/* pippengerMSM — :375-438 */
static aff pippenger_msm(const uint64_t *xy, const uint8_t *inf, const uint64_t *sc, size_t n) {
    size_t c = zo_optimal_window_size(n);
    size_t num_windows = (256 + c - 1) / c;
    size_t num_buckets = ((size_t)1 << c) - 1;
    jac final_result = j_identity();
    jac buckets[256];
    for (size_t w = num_windows; w-- > 0;) {
        if (!f_is_zero(&final_result.z))
            for (size_t i = 0; i < c; i++) final_result = j_double(&final_result);
        for (size_t j = 0; j < num_buckets && j < 256; j++) buckets[j] = j_identity();
        for (size_t i = 0; i < n; i++) {
            if (inf && inf[i]) continue;
            size_t bidx = get_window((const fe *)(sc + 4 * i), w, c);
            if (bidx == 0) continue;
            size_t idx = bidx - 1;
            if (idx < num_buckets) {
                aff b = load_aff(xy, inf, i);
                buckets[idx] = j_add_affine(&buckets[idx], &b);
            }
        }
        jac running = j_identity(), wsum = j_identity();
        for (size_t b = num_buckets; b-- > 0;) {
            running = j_add(&running, &buckets[b]);
            wsum = j_add(&wsum, &running);
        }
        final_result = j_add(&final_result, &wsum);
    }
    return j_to_affine(&final_result);
}
/* MSM.compute — :355-372 */
static aff msm_compute(const uint64_t *xy, const uint8_t *inf, const uint64_t *sc, size_t n) {
    if (n == 0) return a_identity();
    if (n < 8) return naive_msm(xy, inf, sc, n);
    return pippenger_msm(xy, inf, sc, n);
}
EXPORT void zo_msm_g1(const uint64_t *xy, const uint8_t *inf, const uint64_t *scalars, size_t n,
                      uint64_t out_xy[8], uint8_t *out_inf) {
    aff r = msm_compute(xy, inf, scalars, n);
    store_aff(&r, out_xy, out_inf);
}
/* BatchMSM.compute — :545-565 */
EXPORT void zo_msm_g1_batch(const uint64_t *xy, const uint8_t *inf, size_t n,
                            const uint64_t *const *batches, size_t k, uint64_t *out_xy, uint8_t *out_inf) {
    for (size_t b = 0; b < k; b++) {
        aff r = msm_compute(xy, inf, batches[b], n);
        store_aff(&r, out_xy + 8 * b, out_inf ? out_inf + b : NULL);
    }
}

/* ParallelMSM.compute — :572-680: contiguous chunks of ceil(n/T), each
 * SingleMSM.compute -> fromAffine, serial Jacobian combine, toAffine. */
typedef struct { const uint64_t *xy; const uint8_t *inf; const uint64_t *sc; size_t n; jac result; } pm_ctx;
static void *pm_worker(void *p) {
    pm_ctx *c = (pm_ctx *)p;
    if (c->n == 0) { c->result = j_identity(); return NULL; }
    aff a = msm_compute(c->xy, c->inf, c->sc, c->n);
    c->result = j_from_affine(&a);
    return NULL;
}
EXPORT void zo_msm_g1_parallel(const uint64_t *xy, const uint8_t *inf, const uint64_t *scalars, size_t n,
                               size_t num_threads, uint64_t out_xy[8], uint8_t *out_inf) {
    if (n == 0) { aff r = a_identity(); store_aff(&r, out_xy, out_inf); return; }
    size_t by_size = n / 1024; if (by_size < 1) by_size = 1;
    size_t T = num_threads < by_size ? num_threads : by_size;
    if (T <= 1) { zo_msm_g1(xy, inf, scalars, n, out_xy, out_inf); return; }
    size_t chunk = (n + T - 1) / T;
    pm_ctx *ctx = (pm_ctx *)calloc(T, sizeof(pm_ctx));
    pthread_t *th = (pthread_t *)calloc(T, sizeof(pthread_t));
    for (size_t i = 0; i < T; i++) {
        size_t start = i * chunk, end = start + chunk; if (end > n) end = n;
        if (start >= n) { ctx[i].n = 0; }
        else { ctx[i].xy = xy + 8 * start; ctx[i].inf = inf ? inf + start : NULL; ctx[i].sc = scalars + 4 * start; ctx[i].n = end - start; }
        pthread_create(&th[i], NULL, pm_worker, &ctx[i]);
    }
    for (size_t i = 0; i < T; i++) pthread_join(th[i], NULL);
    jac fin = j_identity();
    for (size_t i = 0; i < T; i++) fin = j_add(&fin, &ctx[i].result);
    aff r = j_to_affine(&fin);
    store_aff(&r, out_xy, out_inf);
    free(ctx); free(th);
}
/* MSM.scalarMul(base, scalar).toAffine() */
EXPORT void zo_g1_scalar_mul(const uint64_t xy[8], uint8_t inf, const uint64_t scalar[4], uint64_t out_xy[8], uint8_t *out_inf) {
    aff b = load_aff(xy, &inf, 0);
    jac j = scalar_mul(&b, (const fe *)scalar);
    aff r = j_to_affine(&j);
    store_aff(&r, out_xy, out_inf);
}
EXPORT void zo_g1_add_affine(const uint64_t a[8], uint8_t ainf, const uint64_t b[8], uint8_t binf, uint64_t out_xy[8], uint8_t *out_inf) {
    aff pa = load_aff(a, &ainf, 0), pb = load_aff(b, &binf, 0);
    aff r = a_add(&pa, &pb);
    store_aff(&r, out_xy, out_inf);
}
EXPORT void zo_g1_double_affine(const uint64_t a[8], uint8_t ainf, uint64_t out_xy[8], uint8_t *out_inf) {
    aff pa = load_aff(a, &ainf, 0);
    aff r = a_double(&pa);
    store_aff(&r, out_xy, out_inf);
}
/* Jacobian ops on raw 12-limb records, for unit-testing device formulas */
EXPORT void zo_g1_jac_add(const uint64_t a[12], const uint64_t b[12], uint64_t o[12]) {
    jac r = j_add((const jac *)a, (const jac *)b); memcpy(o, &r, 96);
}
EXPORT void zo_g1_jac_double(const uint64_t a[12], uint64_t o[12]) {
    jac r = j_double((const jac *)a); memcpy(o, &r, 96);
}
EXPORT void zo_g1_jac_add_affine(const uint64_t a[12], const uint64_t b[8], uint8_t binf, uint64_t o[12]) {
    aff pb = load_aff(b, &binf, 0);
    jac r = j_add_affine((const jac *)a, &pb); memcpy(o, &r, 96);
}
EXPORT void zo_g1_jac_to_affine(const uint64_t a[12], uint64_t out_xy[8], uint8_t *out_inf) {
    aff r = j_to_affine((const jac *)a); store_aff(&r, out_xy, out_inf);
}
EXPORT int zo_g1_is_on_curve(const uint64_t xy[8], uint8_t inf) {
    aff p = load_aff(xy, &inf, 0); return a_on_curve(&p);
}
/* Bench point family P_i = (i+1)·G, G=(1,2) — src/bench.zig:261-268. Built by
 * a running Jacobian sum + per-point toAffine replaced with batch inversion
 * (values identical: affine coordinates are unique). */
EXPORT void zo_g1_gen_multiples(size_t n, uint64_t *out_xy) {
    if (n == 0) return;
    const fparams *F = &FP;
    aff g; g.x = f_one(F); g.y = f_from_u64(F, 2); g.inf = 0;
    jac *pts = (jac *)malloc(n * sizeof(jac));
    fe *pre = (fe *)malloc(n * sizeof(fe));
    jac acc = j_from_affine(&g);
    for (size_t i = 0; i < n; i++) { pts[i] = acc; acc = j_add_affine(&acc, &g); }
    fe run = f_one(F);
    for (size_t i = 0; i < n; i++) { pre[i] = run; run = f_mul(F, &run, &pts[i].z); }
    fe inv; f_inv(F, &run, &inv);
    for (size_t i = n; i-- > 0;) {
        fe zi = f_mul(F, &inv, &pre[i]);
        inv = f_mul(F, &inv, &pts[i].z);
        fe zi2 = f_sqr(F, &zi), zi3 = f_mul(F, &zi2, &zi);
        fe x = f_mul(F, &pts[i].x, &zi2), y = f_mul(F, &pts[i].y, &zi3);
        memcpy(out_xy + 8 * i, &x, 32); memcpy(out_xy + 8 * i + 4, &y, 32);
    }
    free(pts); free(pre);
}

/* ---------------------------------------------------------------- HyperKZG */
/* HyperKZG.setup mock SRS: powers[i] = scalarMul(G, tau^i).toAffine(), tau = 0x12345678
 * — src/poly/commitment/mod.zig:174-213 */
EXPORT void zo_hyperkzg_setup(size_t max_degree, uint64_t *out_xy, uint8_t *out_inf) {
    aff g; g.x = f_one(&FP); g.y = f_from_u64(&FP, 2); g.inf = 0;
    fe tau = f_from_u64(&FR, 0x12345678ULL), tp = f_one(&FR);
    for (size_t i = 0; i < max_degree; i++) {
        jac j = scalar_mul(&g, &tp);
        aff a = j_to_affine(&j);
        store_aff(&a, out_xy + 8 * i, out_inf ? out_inf + i : NULL);
        tp = f_mul(&FR, &tp, &tau);
    }
}
/* HyperKZG.commit — :239-255 */
EXPORT void zo_hyperkzg_commit(const uint64_t *srs_xy, const uint8_t *srs_inf, size_t srs_len,
                               const uint64_t *evals, size_t n_evals, uint64_t out_xy[8], uint8_t *out_inf) {
    if (n_evals == 0) { aff r = a_identity(); store_aff(&r, out_xy, out_inf); return; }
    size_t n = n_evals < srs_len ? n_evals : srs_len;
    zo_msm_g1(srs_xy, srs_inf, evals, n, out_xy, out_inf);
}
/* PolyCommitment.toBytes: x||y big-endian canonical; identity -> 64 zero bytes
 * — src/zkvm/commitment_types.zig:49-54 */
EXPORT void zo_commitment_to_bytes(const uint64_t xy[8], uint8_t out64[64]) {
    zo_f_to_bytes_be(1, xy, out64);
    zo_f_to_bytes_be(1, xy + 4, out64 + 32);
}
/* HyperKZG.open — :261-324. quotient commitments (num_vars x 8 limbs + inf) + final eval */
EXPORT void zo_hyperkzg_open(const uint64_t *srs_xy, const uint8_t *srs_inf, size_t srs_len,
                             const uint64_t *evals, size_t n_evals, const uint64_t *point, size_t num_vars,
                             const uint64_t value[4], uint64_t *q_xy, uint8_t *q_inf, uint64_t final_eval[4]) {
    if (num_vars == 0) { memcpy(final_eval, value, 32); return; }
    size_t len = n_evals;
    fe *cur = (fe *)malloc((len ? len : 1) * sizeof(fe));
    memcpy(cur, evals, len * 32);
    for (size_t i = 0; i < num_vars; i++) {
        size_t half = len / 2;
        if (half == 0) break;
        fe *q = (fe *)malloc(half * sizeof(fe));
        for (size_t j = 0; j < half; j++) q[j] = f_sub(&FR, &cur[j + half], &cur[j]);
        zo_hyperkzg_commit(srs_xy, srs_inf, srs_len, (const uint64_t *)q, half, q_xy + 8 * i, q_inf ? q_inf + i : NULL);
        free(q);
        fe *nw = (fe *)malloc(half * sizeof(fe));
        fe one = f_one(&FR), r = *(const fe *)(point + 4 * i), omr = f_sub(&FR, &one, &r);
        for (size_t j = 0; j < half; j++) {
            fe lo = f_mul(&FR, &cur[j], &omr), hi = f_mul(&FR, &cur[j + half], &r);
            nw[j] = f_add(&FR, &lo, &hi);
        }
        free(cur); cur = nw; len = half;
    }
    fe fin = len > 0 ? cur[0] : f_zero();
    memcpy(final_eval, &fin, 32);
    free(cur);
}

/* HyperKZG.evaluateMultilinear — src/poly/commitment/mod.zig:788-817: index bit j <-> point[j] (LSB first); direct
 * monomial sum only when point.len <= 10 and evals.len <= 1024, otherwise the reference's "fallback" returns evals[0]. */
static fe hk_eval_multilinear(const fe *evals, size_t n, const fe *point, size_t v) {
    if (n == 0) return f_zero();
    if (v == 0) return evals[0];
    if (v <= 10 && n <= 1024) {
        fe sum = f_zero(), one = f_one(&FR);
        for (size_t idx = 0; idx < n; idx++) {
            fe term = evals[idx];
            size_t bits = idx;
            for (size_t j = 0; j < v; j++) {
                fe f = (bits & 1) ? point[j] : f_sub(&FR, &one, &point[j]);
                term = f_mul(&FR, &term, &f);
                bits >>= 1;
            }
            sum = f_add(&FR, &sum, &term);
        }
        return sum;
    }
    return evals[0];
}

/* HyperKZG.batchOpen — src/poly/commitment/mod.zig:607-732. polys[i] has lens[i] elements; outputs: up to num_vars
 * quotient commitments (*n_quot computed before the fold runs out of elements, :677-680,714-720), evaluations[k],
 * final_eval, gamma. */
EXPORT void zo_hyperkzg_batch_open(const uint64_t *srs_xy, const uint8_t *srs_inf, size_t srs_len, const uint64_t *const *polys,
                                   const size_t *lens, size_t k, const uint64_t *point, size_t num_vars, uint64_t *q_xy,
                                   uint8_t *q_inf, size_t *n_quot, uint64_t *evaluations, uint64_t final_eval[4], uint64_t gamma_out[4]) {
    *n_quot = 0;
    if (k == 0) {  /* :613-621 */
        fe z = f_zero(), one = f_one(&FR);
        memcpy(final_eval, &z, 32);
        memcpy(gamma_out, &one, 32);
        return;
    }
    const fe *pt = (const fe *)point;
    size_t poly_size = lens[0];
    fe *ev = (fe *)evaluations;
    for (size_t i = 0; i < k; i++) ev[i] = hk_eval_multilinear((const fe *)polys[i], lens[i], pt, num_vars);  /* :628-631 */
    uint64_t g0[4] = {0x9a8b7c6dULL, 0, 0, 0}, e11[4] = {11, 0, 0, 0};
    fe gamma, eleven;
    zo_f_to_mont(0, g0, (uint64_t *)&gamma, 1);   /* F.fromU64 */
    zo_f_to_mont(0, e11, (uint64_t *)&eleven, 1);
    for (size_t j = 0; j < num_vars; j++) {     /* :634-637 */
        fe t = f_add(&FR, &pt[j], &eleven);
        gamma = f_mul(&FR, &gamma, &t);
    }
    fe zero = f_zero();
    if (memcmp(&gamma, &zero, 32) == 0) gamma = f_one(&FR);
    fe *combined = (fe *)malloc((poly_size ? poly_size : 1) * sizeof(fe));
    for (size_t j = 0; j < poly_size; j++) combined[j] = f_zero();
    fe gp = f_one(&FR);
    for (size_t i = 0; i < k; i++) {            /* :646-654 */
        const fe *p = (const fe *)polys[i];
        for (size_t j = 0; j < poly_size; j++)
            if (j < lens[i]) {
                fe t = f_mul(&FR, &gp, &p[j]);
                combined[j] = f_add(&FR, &combined[j], &t);
            }
        gp = f_mul(&FR, &gp, &gamma);
    }
    fe ce = f_zero();
    gp = f_one(&FR);
    for (size_t i = 0; i < k; i++) {            /* :657-662 */
        fe t = f_mul(&FR, &gp, &ev[i]);
        ce = f_add(&FR, &ce, &t);
        gp = f_mul(&FR, &gp, &gamma);
    }
    memcpy(gamma_out, &gamma, 32);
    if (num_vars == 0) {                        /* :665-673 */
        memcpy(final_eval, &ce, 32);
        free(combined);
        return;
    }
    size_t len = poly_size;
    fe *cur = combined;
    for (size_t i = 0; i < num_vars; i++) {     /* :684-712 */
        size_t half = len / 2;
        if (half == 0) break;
        fe *q = (fe *)malloc(half * sizeof(fe));
        for (size_t j = 0; j < half; j++) q[j] = f_sub(&FR, &cur[j + half], &cur[j]);
        zo_hyperkzg_commit(srs_xy, srs_inf, srs_len, (const uint64_t *)q, half, q_xy + 8 * i, q_inf ? q_inf + i : NULL);
        (*n_quot)++;
        free(q);
        fe *nw = (fe *)malloc(half * sizeof(fe));
        fe one = f_one(&FR), r = pt[i], omr = f_sub(&FR, &one, &r);
        for (size_t j = 0; j < half; j++) {
            fe lo = f_mul(&FR, &cur[j], &omr), hi = f_mul(&FR, &cur[j + half], &r);
            nw[j] = f_add(&FR, &lo, &hi);
        }
        free(cur); cur = nw; len = half;
    }
    fe fin = len > 0 ? cur[0] : f_zero();       /* :714 */
    memcpy(final_eval, &fin, 32);
    free(cur);
}

/* -------------------------------------------------------------------- poly */
/* EqPolynomial.evalsSliceWithScaling — src/poly/mod.zig:252-290 (big-endian index) */
EXPORT void zo_fr_eq_table(const uint64_t *r, size_t v, const uint64_t *scale, uint64_t *out) {
    fe *res = (fe *)out;
    size_t final_size = (size_t)1 << v;
    for (size_t i = 0; i < final_size; i++) res[i] = f_zero();
    res[0] = scale ? *(const fe *)scale : f_one(&FR);
    size_t size = 1;
    for (size_t j = v; j-- > 0;) {
        const fe *rj = (const fe *)(r + 4 * j);
        for (size_t i = 0; i < size; i++) {
            fe x = res[i], y = f_mul(&FR, &x, rj);
            res[i + size] = y;
            res[i] = f_sub(&FR, &x, &y);
        }
        size *= 2;
    }
}
/* EqPolynomial.mle / EqPolynomial.evaluate — src/poly/mod.zig:311-321 (and :214-227, the same product):
 * prod_i (r_i*x_i + (1-r_i)*(1-x_i)) */
EXPORT void zo_fr_eq_mle(const uint64_t *r, const uint64_t *x, size_t v, uint64_t out[4]) {
    fe one = f_one(&FR), result = one;
    for (size_t i = 0; i < v; i++) {
        const fe *ri = (const fe *)(r + 4 * i), *xi = (const fe *)(x + 4 * i);
        fe ri_xi = f_mul(&FR, ri, xi);
        fe one_minus_ri = f_sub(&FR, &one, ri), one_minus_xi = f_sub(&FR, &one, xi);
        fe prod = f_mul(&FR, &one_minus_ri, &one_minus_xi);
        fe term = f_add(&FR, &ri_xi, &prod);
        result = f_mul(&FR, &result, &term);
    }
    memcpy(out, &result, 32);
}
/* DensePolynomial.add / scale — src/poly/mod.zig:94-126 */
EXPORT void zo_fr_poly_scale(const uint64_t *a, size_t n, const uint64_t s[4], uint64_t *out) {
    const fe *sv = (const fe *)s;
    for (size_t i = 0; i < n; i++) {
        fe v = f_mul(&FR, (const fe *)(a + 4 * i), sv);
        memcpy(out + 4 * i, &v, 32);
    }
}
/* GruenSplitEqPolynomial prefix-table step ("append LSB") — src/poly/split_eq.zig:122-147:
 * next[2i] = prev[i]*(1-tau_k), next[2i+1] = prev[i]*tau_k; full table for tau[0..v) */
EXPORT void zo_fr_eq_table_append_lsb(const uint64_t *tau, size_t v, uint64_t *out) {
    size_t total = (size_t)1 << v;
    fe *a = (fe *)malloc(total * sizeof(fe)), *b = (fe *)malloc(total * sizeof(fe));
    a[0] = f_one(&FR);
    for (size_t k = 0; k < v; k++) {
        size_t prev = (size_t)1 << k;
        fe one = f_one(&FR), tk = *(const fe *)(tau + 4 * k), omt = f_sub(&FR, &one, &tk);
        for (size_t i = 0; i < prev; i++) {
            b[2 * i] = f_mul(&FR, &a[i], &omt);
            b[2 * i + 1] = f_mul(&FR, &a[i], &tk);
        }
        fe *t = a; a = b; b = t;
    }
    memcpy(out, a, total * 32);
    free(a); free(b);
}
/* GruenSplitEqPolynomial.initWithScaling's table set — src/poly/split_eq.zig:122-171: E_vec[0] = [1], E_vec[k+1] built from
 * E_vec[k] by the append-LSB step above, every level kept. out: the v+1 tables back to back, table k at element 2^k - 1. */
EXPORT void zo_fr_eq_prefix_tables(const uint64_t *tau, size_t v, uint64_t *out) {
    fe *o = (fe *)out;
    o[0] = f_one(&FR);
    for (size_t k = 0; k < v; k++) {
        size_t prev_size = (size_t)1 << k;
        const fe *prev = o + (prev_size - 1);
        fe *next = o + (2 * prev_size - 1);
        fe one = f_one(&FR), tk = *(const fe *)(tau + 4 * k), omt = f_sub(&FR, &one, &tk);
        for (size_t i = 0; i < prev_size; i++) {
            next[2 * i] = f_mul(&FR, &prev[i], &omt);
            next[2 * i + 1] = f_mul(&FR, &prev[i], &tk);
        }
    }
}
/* GruenSplitEqPolynomial.bind's scalar update — :213-219: current_scalar *= tau_i*r + (1-tau_i)*(1-r) */
EXPORT void zo_gruen_bind_scalar(const uint64_t cur[4], const uint64_t tau_i[4], const uint64_t r[4], uint64_t out[4]) {
    fe one = f_one(&FR), t = *(const fe *)tau_i, rv = *(const fe *)r;
    fe a = f_mul(&FR, &t, &rv), omt = f_sub(&FR, &one, &t), omr = f_sub(&FR, &one, &rv), b = f_mul(&FR, &omt, &omr);
    fe e = f_add(&FR, &a, &b), res = f_mul(&FR, (const fe *)cur, &e);
    memcpy(out, &res, 32);
}
/* GruenSplitEqPolynomial.computeCubicRoundPoly — :353-434 (current_index > 0): l(X) from (current_scalar, tau_curr),
 * q(1) = (claim - l(0) q(0)) / l(1) (zero when l(1) = 0), q(2), q(3) by the recurrences, s = l*q at 0..3 */
EXPORT void zo_gruen_cubic_round_poly(const uint64_t current_scalar[4], const uint64_t tau_curr[4], const uint64_t q_constant[4],
                                      const uint64_t q_quadratic[4], const uint64_t previous_claim[4], uint64_t out[16]) {
    fe one = f_one(&FR), cs = *(const fe *)current_scalar, tc = *(const fe *)tau_curr;
    fe c = *(const fe *)q_constant, e = *(const fe *)q_quadratic, claim = *(const fe *)previous_claim;
    fe omt = f_sub(&FR, &one, &tc), eq0 = f_mul(&FR, &cs, &omt), eq1 = f_mul(&FR, &cs, &tc), slope = f_sub(&FR, &eq1, &eq0);
    fe two = f_from_u64(&FR, 2), three = f_from_u64(&FR, 3);
    fe t2 = f_mul(&FR, &slope, &two), t3 = f_mul(&FR, &slope, &three), l2 = f_add(&FR, &eq0, &t2), l3 = f_add(&FR, &eq0, &t3);
    fe l0q0 = f_mul(&FR, &eq0, &c), q1 = f_zero(), inv;
    if (f_inv(&FR, &eq1, &inv)) { fe d = f_sub(&FR, &claim, &l0q0); q1 = f_mul(&FR, &d, &inv); }
    fe e2 = f_add(&FR, &e, &e);
    fe q2 = f_add(&FR, &q1, &q1); q2 = f_sub(&FR, &q2, &c); q2 = f_add(&FR, &q2, &e2);
    fe q3 = f_add(&FR, &q2, &q1); q3 = f_sub(&FR, &q3, &c); q3 = f_add(&FR, &q3, &e2); q3 = f_add(&FR, &q3, &e2);
    fe s[4] = {l0q0, f_mul(&FR, &eq1, &q1), f_mul(&FR, &l2, &q2), f_mul(&FR, &l3, &q3)};
    memcpy(out, s, 128);
}
/* DensePolynomial.bindLow (in place) — src/poly/mod.zig:160-175 */
EXPORT void zo_fr_bind_low(uint64_t *table, size_t len, const uint64_t r[4]) {
    fe *t = (fe *)table; const fe *rv = (const fe *)r;
    size_t ns = len / 2;
    for (size_t i = 0; i < ns; i++) {
        fe lo = t[2 * i], hi = t[2 * i + 1];
        fe d = f_sub(&FR, &hi, &lo), m = f_mul(&FR, rv, &d);
        t[i] = f_add(&FR, &lo, &m);
    }
}
/* DensePolynomial.bindFirst (high half, new array) — :128-149 */
EXPORT void zo_fr_bind_high(const uint64_t *table, size_t len, const uint64_t r[4], uint64_t *out) {
    const fe *t = (const fe *)table; fe *o = (fe *)out; const fe *rv = (const fe *)r;
    size_t ns = len / 2;
    fe one = f_one(&FR), omr = f_sub(&FR, &one, rv);
    for (size_t i = 0; i < ns; i++) {
        fe lo = f_mul(&FR, &t[i], &omr), hi = f_mul(&FR, &t[i + ns], rv);
        o[i] = f_add(&FR, &lo, &hi);
    }
}
/* JoltSpartanInterface.bindChallenge fold: new[i] = (1-r)*old[2i] + r*old[2i+1]
 * — src/zkvm/r1cs/jolt_r1cs.zig:470-477 (out may alias table) */
EXPORT void zo_fr_bind_low_2mul(const uint64_t *table, size_t len, const uint64_t r[4], uint64_t *out) {
    const fe *t = (const fe *)table; fe *o = (fe *)out; const fe *rv = (const fe *)r;
    size_t ns = len / 2;
    fe one = f_one(&FR), omr = f_sub(&FR, &one, rv);
    for (size_t i = 0; i < ns; i++) {
        fe lo = f_mul(&FR, &omr, &t[2 * i]), hi = f_mul(&FR, rv, &t[2 * i + 1]);
        o[i] = f_add(&FR, &lo, &hi);
    }
}
/* DensePolynomial.evaluate — :73-92 (index bit j <-> point[j], LSB-first). O(n·v): small n only */
EXPORT void zo_fr_dense_evaluate(const uint64_t *evals, size_t num_vars, const uint64_t *point, uint64_t out[4]) {
    fe result = f_zero(), one = f_one(&FR);
    size_t n = (size_t)1 << num_vars;
    for (size_t i = 0; i < n; i++) {
        fe term = *(const fe *)(evals + 4 * i);
        for (size_t j = 0; j < num_vars; j++) {
            const fe *pj = (const fe *)(point + 4 * j);
            if ((i >> j) & 1) term = f_mul(&FR, &term, pj);
            else { fe om = f_sub(&FR, &one, pj); term = f_mul(&FR, &term, &om); }
        }
        result = f_add(&FR, &result, &term);
    }
    memcpy(out, &result, 32);
}
/* Sumcheck.Prover.nextRound sums: g0 = Σ first half, g1 = Σ second half
 * — src/subprotocols/mod.zig:79-93 */
EXPORT void zo_fr_sum_halves(const uint64_t *table, size_t len, uint64_t g0[4], uint64_t g1[4]) {
    const fe *t = (const fe *)table; size_t half = len / 2;
    fe a = f_zero(), b = f_zero();
    for (size_t i = 0; i < half; i++) a = f_add(&FR, &a, &t[i]);
    for (size_t i = 0; i < half; i++) b = f_add(&FR, &b, &t[i + half]);
    memcpy(g0, &a, 32); memcpy(g1, &b, 32);
}
/* even/odd sums used by the LowToHigh provers — src/zkvm/r1cs/jolt_r1cs.zig:436-444 */
EXPORT void zo_fr_sum_even_odd(const uint64_t *table, size_t len, uint64_t s0[4], uint64_t s1[4]) {
    const fe *t = (const fe *)table; size_t half = len / 2;
    fe a = f_zero(), b = f_zero();
    for (size_t i = 0; i < half; i++) { a = f_add(&FR, &a, &t[2 * i]); b = f_add(&FR, &b, &t[2 * i + 1]); }
    memcpy(s0, &a, 32); memcpy(s1, &b, 32);
}
/* Spartan combine f[i] = eq[i]·(Az[i]·Bz[i] − Cz[i]) — src/zkvm/spartan/mod.zig:191-199 */
EXPORT void zo_fr_spartan_combine(const uint64_t *eq, const uint64_t *az, const uint64_t *bz, const uint64_t *cz,
                                  size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; i++) {
        fe ab = f_mul(&FR, (const fe *)(az + 4 * i), (const fe *)(bz + 4 * i));
        fe d = f_sub(&FR, &ab, (const fe *)(cz + 4 * i));
        fe f = f_mul(&FR, (const fe *)(eq + 4 * i), &d);
        memcpy(out + 4 * i, &f, 32);
    }
}
/* UniPoly.evaluate (Horner) — src/poly/mod.zig:608-618 */
static fe unipoly_eval(const fe *coeffs, size_t n, const fe *x) {
    if (n == 0) return f_zero();
    fe result = coeffs[n - 1];
    for (size_t i = n - 1; i-- > 0;) { result = f_mul(&FR, &result, x); result = f_add(&FR, &result, &coeffs[i]); }
    return result;
}
/* Verifier.deriveChallenge toy mixer — src/subprotocols/mod.zig:211-243 */
static fe derive_challenge(size_t round, const fe *claim, const fe *coeffs, size_t ncoeffs) {
    uint64_t h = 0x9e3779b97f4a7c15ULL;
    h ^= (uint64_t)round; h *= 0xff51afd7ed558ccdULL;
    for (int i = 0; i < 4; i++) { h ^= claim->l[i]; h *= 0xc4ceb9fe1a85ec53ULL; }
    for (size_t c = 0; c < ncoeffs; c++)
        for (int i = 0; i < 4; i++) { h ^= coeffs[c].l[i]; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; }
    h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33;
    return f_from_u64(&FR, h);
}
EXPORT void zo_sumcheck_derive_challenge(size_t round, const uint64_t claim[4], const uint64_t *coeffs, size_t ncoeffs, uint64_t out[4]) {
    fe r = derive_challenge(round, (const fe *)claim, (const fe *)coeffs, ncoeffs); memcpy(out, &r, 32);
}
/* runSumcheck — src/subprotocols/mod.zig:302-354, with Prover.nextRound (:69-109),
 * Verifier.verifyRound (:165-207), Prover.receiveChallenge (:112-122).
 * outputs: claim[4]; rounds[num_vars][2][4] = [g0, g1-g0]; challenges[num_vars][4];
 * final_eval[4]; returns 1 if verifier.claim == final_eval, 0 if not, -1 on
 * SumcheckVerificationFailed. */
EXPORT int zo_run_sumcheck(const uint64_t *evals, size_t num_vars, uint64_t claim_out[4], uint64_t *rounds,
                           uint64_t *challenges, uint64_t final_eval[4]) {
    size_t len = (size_t)1 << num_vars;
    fe *cur = (fe *)malloc(len * sizeof(fe));
    memcpy(cur, evals, len * 32);
    fe claim = f_zero();
    for (size_t i = 0; i < len; i++) claim = f_add(&FR, &claim, &cur[i]);
    memcpy(claim_out, &claim, 32);
    fe vclaim = claim;
    for (size_t rd = 0; rd < num_vars; rd++) {
        fe coeffs[2], g0, g1;
        zo_fr_sum_halves((const uint64_t *)cur, len, g0.l, g1.l);
        coeffs[0] = g0; coeffs[1] = f_sub(&FR, &g1, &g0);
        memcpy(rounds + 8 * rd, coeffs, 64);
        fe zero = f_zero(), one = f_one(&FR);
        fe p0 = unipoly_eval(coeffs, 2, &zero), p1 = unipoly_eval(coeffs, 2, &one);
        fe sum = f_add(&FR, &p0, &p1);
        if (!f_eq(&sum, &vclaim)) { free(cur); return -1; }
        fe ch = derive_challenge(rd, &vclaim, coeffs, 2);
        memcpy(challenges + 4 * rd, &ch, 32);
        vclaim = unipoly_eval(coeffs, 2, &ch);
        fe *nw = (fe *)malloc((len / 2) * sizeof(fe));
        zo_fr_bind_high((const uint64_t *)cur, len, ch.l, (uint64_t *)nw);
        free(cur); cur = nw; len /= 2;
    }
    memcpy(final_eval, &cur[0], 32);
    int ok = f_eq(&vclaim, &cur[0]);
    free(cur);
    return ok;
}

/* ================================================================== prover fold sites + host transcript
 * Keccak Fiat-Shamir transcript — src/transcripts/mod.zig:19-221. State = 200 bytes + a position; bytes are XORed in at
 * `position`, a Keccak-f[1600] every 136 bytes (no padding); challengeScalar = append label, one Keccak-f,
 * F.fromBytes(state[0..32]) (:116-130). appendScalar absorbs the element's raw Montgomery limbs, little-endian (:100-110). */
static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const unsigned KECCAK_ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const unsigned KECCAK_PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
static inline uint64_t rotl64(uint64_t x, unsigned n) { return (x << n) | (x >> (64 - n)); }

/* keccakF — :163-213, on 25 little-endian lanes */
EXPORT void zo_keccak_f1600(uint64_t st[25]) {
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            uint64_t t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
            for (int j = i; j < 25; j += 5) st[j] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; i++) {
            unsigned j = KECCAK_PILN[i];
            uint64_t tmp = st[j];
            st[j] = rotl64(t, KECCAK_ROTC[i]);
            t = tmp;
        }
        for (int j = 0; j < 5; j++) {
            int row = j * 5;
            for (int i = 0; i < 5; i++) bc[i] = st[row + i];
            for (int i = 0; i < 5; i++) st[row + i] = bc[i] ^ (~bc[(i + 1) % 5] & bc[(i + 2) % 5]);
        }
        st[0] ^= KECCAK_RC[round];
    }
}

typedef struct { uint8_t state[200]; uint64_t position; } zo_transcript;  /* 208 bytes, plain data: callers keep it in a byte buffer */

static void tr_permute(zo_transcript *t) {
    uint64_t st[25];
    for (int i = 0; i < 25; i++) { uint64_t v = 0; for (int b = 7; b >= 0; b--) v = (v << 8) | t->state[8 * i + b]; st[i] = v; }
    zo_keccak_f1600(st);
    for (int i = 0; i < 25; i++) for (int b = 0; b < 8; b++) t->state[8 * i + b] = (uint8_t)(st[i] >> (8 * b));
}
/* appendBytes — :88-98 */
EXPORT void zo_transcript_append_bytes(zo_transcript *t, const uint8_t *data, size_t n) {
    for (size_t i = 0; i < n; i++) {
        t->state[t->position] ^= data[i];
        t->position += 1;
        if (t->position >= 136) { tr_permute(t); t->position = 0; }
    }
}
/* init — :61-74 */
EXPORT void zo_transcript_init(zo_transcript *t, const uint8_t *domain, size_t n) {
    memset(t, 0, sizeof(*t));
    zo_transcript_append_bytes(t, domain, n);
}
/* appendScalar — :100-110 */
EXPORT void zo_transcript_append_scalar(zo_transcript *t, const uint8_t *label, size_t label_len, const uint64_t scalar[4]) {
    zo_transcript_append_bytes(t, label, label_len);
    uint8_t buf[32];
    for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) buf[8 * i + b] = (uint8_t)(scalar[i] >> (8 * b));
    zo_transcript_append_bytes(t, buf, 32);
}
/* challengeScalar — :116-130; F.fromBytes — src/field/mod.zig:625-639 (raw little-endian limbs times R^2) */
EXPORT void zo_transcript_challenge_scalar(zo_transcript *t, const uint8_t *label, size_t label_len, uint64_t out[4]) {
    zo_transcript_append_bytes(t, label, label_len);
    tr_permute(t);
    fe raw;
    for (int i = 0; i < 4; i++) { uint64_t v = 0; for (int b = 7; b >= 0; b--) v = (v << 8) | t->state[8 * i + b]; raw.l[i] = v; }
    fe r = f_to_mont(&FR, &raw);
    memcpy(out, &r, 32);
}

/* Stage 1 (outer Spartan sumcheck) round loop — src/zkvm/prover.zig:397-432 over JoltSpartanInterface.computeRoundPolynomial /
 * bindChallenge (src/zkvm/r1cs/jolt_r1cs.zig:413-486): per round [p0, p1, p2 = 2 p1 - p0] with p0 / p1 the sums over even /
 * odd indices (current_len <= 1: [poly[0] or 0, 0, 0] and no fold); the three values are absorbed as "round_poly_0/1/2", the
 * challenge is challengeScalar("spartan_round"), the fold is new[i] = (1-r) old[2i] + r old[2i+1]. combined: len entries
 * (modified in place); outputs: round_polys num_rounds x 3 x 4, challenges num_rounds x 4, final_eval = combined_poly[0]. */
EXPORT void zo_stage1_prove(uint64_t *combined, size_t len, size_t num_rounds, zo_transcript *t, uint64_t *round_polys,
                            uint64_t *challenges, uint64_t final_eval[4]) {
    fe *poly = (fe *)combined;
    size_t cur = len;
    for (size_t round = 0; round < num_rounds; round++) {
        fe p0 = f_zero(), p1 = f_zero(), p2 = f_zero();
        if (cur <= 1) {
            if (cur == 1) p0 = poly[0];
        } else {
            size_t half = cur / 2;
            for (size_t i = 0; i < half; i++) { p0 = f_add(&FR, &p0, &poly[2 * i]); p1 = f_add(&FR, &p1, &poly[2 * i + 1]); }
            fe d = f_add(&FR, &p1, &p1);
            p2 = f_sub(&FR, &d, &p0);
        }
        memcpy(round_polys + 12 * round, &p0, 32); memcpy(round_polys + 12 * round + 4, &p1, 32); memcpy(round_polys + 12 * round + 8, &p2, 32);
        zo_transcript_append_scalar(t, (const uint8_t *)"round_poly_0", 12, p0.l);
        zo_transcript_append_scalar(t, (const uint8_t *)"round_poly_1", 12, p1.l);
        zo_transcript_append_scalar(t, (const uint8_t *)"round_poly_2", 12, p2.l);
        fe ch;
        zo_transcript_challenge_scalar(t, (const uint8_t *)"spartan_round", 13, ch.l);
        memcpy(challenges + 4 * round, &ch, 32);
        if (cur > 1) {
            size_t half = cur / 2;
            fe one = f_one(&FR), omr = f_sub(&FR, &one, &ch);
            for (size_t i = 0; i < half; i++) {
                fe a = f_mul(&FR, &omr, &poly[2 * i]), b = f_mul(&FR, &ch, &poly[2 * i + 1]);
                poly[i] = f_add(&FR, &a, &b);
            }
            cur = half;
        }
    }
    fe fin = len ? poly[0] : f_zero();
    memcpy(final_eval, &fin, 32);
}

/* RafEvaluationProver.computeRoundPolynomialCubic — src/zkvm/ram/raf_checking.zig:335-410. ra: the 2^num_vars active entries
 * (LowToHigh pairs 2i, 2i+1); bound: the `round` challenges bound so far; unmap_num_vars = UnmapPolynomial.num_vars.
 * out: s(0), s(1), s(2), s(3). */
EXPORT void zo_raf_round_cubic(const uint64_t *ra, size_t ra_num_vars, uint64_t start_address, const uint64_t *bound, size_t round,
                               size_t unmap_num_vars, const uint64_t current_claim[4], uint64_t out[16]) {
    const fe *evals = (const fe *)ra;
    size_t active_len = (size_t)1 << ra_num_vars, half = active_len / 2;
    fe s0 = f_zero(), s2 = f_zero();
    fe base = f_from_u64(&FR, start_address);
    uint64_t power = 8;
    for (size_t j = 0; j < round; j++) {
        fe pw = f_from_u64(&FR, power), term = f_mul(&FR, (const fe *)(bound + 4 * j), &pw);
        base = f_add(&FR, &base, &term);
        power *= 2;
    }
    const uint64_t current_power = power;
    for (size_t i = 0; i < half; i++) {
        fe ra_lo = evals[2 * i], ra_hi = evals[2 * i + 1];
        fe dbl = f_add(&FR, &ra_hi, &ra_hi), ra_at_2 = f_sub(&FR, &dbl, &ra_lo);
        fe remaining = f_zero();
        uint64_t remaining_power = current_power * 2;
        size_t remaining_vars = unmap_num_vars - round - 1, idx = i;
        for (size_t k = 0; k < remaining_vars; k++) {
            if (idx & 1) { fe rp = f_from_u64(&FR, remaining_power); remaining = f_add(&FR, &remaining, &rp); }
            idx >>= 1;
            remaining_power *= 2;
        }
        fe u0 = f_add(&FR, &base, &remaining);
        fe cp2 = f_from_u64(&FR, current_power * 2), u2 = f_add(&FR, &base, &cp2);
        u2 = f_add(&FR, &u2, &remaining);
        fe t0 = f_mul(&FR, &ra_lo, &u0), t2 = f_mul(&FR, &ra_at_2, &u2);
        s0 = f_add(&FR, &s0, &t0);
        s2 = f_add(&FR, &s2, &t2);
    }
    fe s1 = f_sub(&FR, (const fe *)current_claim, &s0);
    fe three = f_from_u64(&FR, 3), a = f_mul(&FR, &s1, &three), b = f_mul(&FR, &s2, &three);
    fe s3 = f_sub(&FR, &s0, &a);
    s3 = f_add(&FR, &s3, &b);
    memcpy(out, &s0, 32); memcpy(out + 4, &s1, 32); memcpy(out + 8, &s2, 32); memcpy(out + 12, &s3, 32);
}

/* RafEvaluationProver.updateClaim — :420-445: Lagrange interpolation of evals at 0,1,2,3 at the challenge */
EXPORT void zo_raf_update_claim(const uint64_t evals[16], const uint64_t challenge[4], uint64_t out[4]) {
    const fe *e = (const fe *)evals;
    fe c = *(const fe *)challenge, one = f_one(&FR), zero = f_zero();
    fe two = f_from_u64(&FR, 2), three = f_from_u64(&FR, 3), six = f_from_u64(&FR, 6);
    fe cm1 = f_sub(&FR, &c, &one), cm2 = f_sub(&FR, &c, &two), cm3 = f_sub(&FR, &c, &three);
    fe neg6 = f_sub(&FR, &zero, &six), neg2 = f_sub(&FR, &zero, &two);
    fe i_neg6, i_2, i_neg2, i_6;  /* .inverse().? — none of -6, 2, -2, 6 is zero */
    (void)f_inv(&FR, &neg6, &i_neg6); (void)f_inv(&FR, &two, &i_2); (void)f_inv(&FR, &neg2, &i_neg2); (void)f_inv(&FR, &six, &i_6);
    fe L0 = f_mul(&FR, &cm1, &cm2); L0 = f_mul(&FR, &L0, &cm3); L0 = f_mul(&FR, &L0, &i_neg6);
    fe L1 = f_mul(&FR, &c, &cm2); L1 = f_mul(&FR, &L1, &cm3); L1 = f_mul(&FR, &L1, &i_2);
    fe L2 = f_mul(&FR, &c, &cm1); L2 = f_mul(&FR, &L2, &cm3); L2 = f_mul(&FR, &L2, &i_neg2);
    fe L3 = f_mul(&FR, &c, &cm1); L3 = f_mul(&FR, &L3, &cm2); L3 = f_mul(&FR, &L3, &i_6);
    fe r = f_mul(&FR, &e[0], &L0), t = f_mul(&FR, &e[1], &L1);
    r = f_add(&FR, &r, &t);
    t = f_mul(&FR, &e[2], &L2); r = f_add(&FR, &r, &t);
    t = f_mul(&FR, &e[3], &L3); r = f_add(&FR, &r, &t);
    memcpy(out, &r, 32);
}

/* LassoProver.computeAddressRoundPoly's two sums — src/zkvm/lasso/prover.zig:283-293: eq_evals split by bit `round` of the
 * u128 lookup index (idx: n x 2 u64, little-endian halves). */
EXPORT void zo_lasso_address_sums(const uint64_t *eq_evals, const uint64_t *idx, size_t n, unsigned round_bit, uint64_t sum0[4],
                                  uint64_t sum1[4]) {
    fe s0 = f_zero(), s1 = f_zero();
    for (size_t j = 0; j < n; j++) {
        uint64_t word = round_bit < 64 ? idx[2 * j] : idx[2 * j + 1];
        unsigned bit = (unsigned)((word >> (round_bit & 63)) & 1);
        const fe *v = (const fe *)(eq_evals + 4 * j);
        if (bit == 0) s0 = f_add(&FR, &s0, v); else s1 = f_add(&FR, &s1, v);
    }
    memcpy(sum0, &s0, 32); memcpy(sum1, &s1, 32);
}

/* SplitEqPolynomial.buildTables + getEq (src/zkvm/lasso/split_eq.zig:113-142,157-168) and LassoProver.init's eq_evals
 * (src/zkvm/lasso/prover.zig:153-171): E_out / E_in built LSB-first (w[i] <-> bit i of the half's index),
 * eq_evals[j] = E_out[j >> num_inner] * E_in[j & mask] for j < num_cycles, zero up to 2^(num_outer + num_inner);
 * *claim = sum of all entries (:166-171). */
EXPORT void zo_lasso_init_eq_evals(const uint64_t *w, size_t num_outer, size_t num_inner, size_t num_cycles, uint64_t *eq_evals,
                                   uint64_t claim[4]) {
    size_t so = (size_t)1 << num_outer, si = (size_t)1 << num_inner, padded = so * si;
    fe *eo = (fe *)malloc(so * sizeof(fe)), *ei = (fe *)malloc(si * sizeof(fe));
    fe one = f_one(&FR);
    eo[0] = one;
    for (size_t i = 0; i < num_outer; i++) {
        size_t half = (size_t)1 << i;
        fe val = *(const fe *)(w + 4 * i), om = f_sub(&FR, &one, &val);
        for (size_t j = 0; j < half; j++) { eo[j + half] = f_mul(&FR, &eo[j], &val); eo[j] = f_mul(&FR, &eo[j], &om); }
    }
    ei[0] = one;
    for (size_t i = 0; i < num_inner; i++) {
        size_t half = (size_t)1 << i;
        fe val = *(const fe *)(w + 4 * (num_outer + i)), om = f_sub(&FR, &one, &val);
        for (size_t j = 0; j < half; j++) { ei[j + half] = f_mul(&FR, &ei[j], &val); ei[j] = f_mul(&FR, &ei[j], &om); }
    }
    fe sum = f_zero(), *out = (fe *)eq_evals;
    for (size_t j = 0; j < padded; j++) {
        out[j] = j < num_cycles ? f_mul(&FR, &eo[j >> num_inner], &ei[j & (si - 1)]) : f_zero();
        sum = f_add(&FR, &sum, &out[j]);
    }
    memcpy(claim, &sum, 32);
    free(eo); free(ei);
}
/* LassoProver.receiveChallenge, address branch — src/zkvm/lasso/prover.zig:375-399: the n lookups' eq values are multiplied by
 * the challenge or its complement by bit `round_bit` of the index; the new claim sums the whole (padded) array. */
EXPORT void zo_lasso_receive_address(uint64_t *eq_evals, size_t padded, const uint64_t *idx, size_t n, unsigned round_bit,
                                     const uint64_t challenge[4], uint64_t claim[4]) {
    fe *e = (fe *)eq_evals, c = *(const fe *)challenge, one = f_one(&FR), omr = f_sub(&FR, &one, &c);
    for (size_t j = 0; j < n; j++) {
        uint64_t word = round_bit < 64 ? idx[2 * j] : idx[2 * j + 1];
        e[j] = ((word >> (round_bit & 63)) & 1) ? f_mul(&FR, &e[j], &c) : f_mul(&FR, &e[j], &omr);
    }
    fe sum = f_zero();
    for (size_t j = 0; j < padded; j++) sum = f_add(&FR, &sum, &e[j]);
    memcpy(claim, &sum, 32);
}
/* deriveChallenge — src/zkvm/lasso/prover.zig:533-551 */
EXPORT void zo_lasso_derive_challenge(const uint64_t *coeffs, size_t ncoeffs, uint64_t round, uint64_t out[4]) {
    uint64_t hash = 0x9e3779b97f4a7c15ULL;
    hash ^= round;
    hash *= 0xff51afd7ed558ccdULL;
    for (size_t i = 0; i < ncoeffs; i++)
        for (int l = 0; l < 4; l++) { hash ^= coeffs[4 * i + l]; hash *= 0xc4ceb9fe1a85ec53ULL; }
    hash ^= hash >> 33;
    fe r = f_from_u64(&FR, hash);
    memcpy(out, &r, 32);
}

/* ---------------------------------------------------------- product-form prover rounds (zkvm) */
/* UniPoly.interpolateDegree3 / evalsToCompressed — src/poly/mod.zig:632-685: [c0, c1, c2, c3] from p(0..3); compressed = [c0, c2, c3] */
EXPORT void zo_interpolate_degree3(const uint64_t evals[16], uint64_t coeffs[16]) {
    const fe *p = (const fe *)evals;
    fe six = f_from_u64(&FR, 6), two = f_from_u64(&FR, 2), inv6, inv2, zero = f_zero();
    (void)f_inv(&FR, &six, &inv6); (void)f_inv(&FR, &two, &inv2);
    fe k11 = f_from_u64(&FR, 11), k18 = f_from_u64(&FR, 18), k9 = f_from_u64(&FR, 9), k5 = f_from_u64(&FR, 5), k4 = f_from_u64(&FR, 4),
       k3 = f_from_u64(&FR, 3);
    fe t, c1 = zero, c2, c3;
    t = f_mul(&FR, &k11, &p[0]); c1 = f_sub(&FR, &c1, &t);
    t = f_mul(&FR, &k18, &p[1]); c1 = f_add(&FR, &c1, &t);
    t = f_mul(&FR, &k9, &p[2]); c1 = f_sub(&FR, &c1, &t);
    t = f_mul(&FR, &two, &p[3]); c1 = f_add(&FR, &c1, &t);
    c1 = f_mul(&FR, &c1, &inv6);
    c2 = f_mul(&FR, &two, &p[0]);
    t = f_mul(&FR, &k5, &p[1]); c2 = f_sub(&FR, &c2, &t);
    t = f_mul(&FR, &k4, &p[2]); c2 = f_add(&FR, &c2, &t);
    c2 = f_sub(&FR, &c2, &p[3]);
    c2 = f_mul(&FR, &c2, &inv2);
    c3 = f_sub(&FR, &zero, &p[0]);
    t = f_mul(&FR, &k3, &p[1]); c3 = f_add(&FR, &c3, &t);
    t = f_mul(&FR, &k3, &p[2]); c3 = f_sub(&FR, &c3, &t);
    c3 = f_add(&FR, &c3, &p[3]);
    c3 = f_mul(&FR, &c3, &inv6);
    fe out[4] = {p[0], c1, c2, c3};
    memcpy(coeffs, out, 128);
}
/* ValEvaluationProver.computeRoundPolynomial — src/zkvm/ram/val_evaluation.zig:554-603 (n = effectiveLen; lt == NULL: the two-table
 * form of ValFinalProver.computeRoundPolynomial, src/zkvm/ram/val_final.zig:149-185) */
EXPORT void zo_val_evaluation_round(const uint64_t *inc, const uint64_t *wa, const uint64_t *lt, size_t n, uint64_t evals[16]) {
    fe e[4] = {f_zero(), f_zero(), f_zero(), f_zero()};
    const fe *I = (const fe *)inc, *W = (const fe *)wa, *L = (const fe *)lt;
    size_t half = n / 2;
    if (half == 0) {
        if (n > 0) { e[0] = f_mul(&FR, &I[0], &W[0]); if (L) e[0] = f_mul(&FR, &e[0], &L[0]); }
        memcpy(evals, e, 128);
        return;
    }
    fe two = f_from_u64(&FR, 2), three = f_from_u64(&FR, 3);
    for (size_t i = 0; i < half; i++) {
        const fe *t[3] = {I, W, L};
        fe v[4];
        for (int k = 0; k < (L ? 3 : 2); k++) {
            fe f0 = t[k][2 * i], f1 = t[k][2 * i + 1];
            fe a = f_mul(&FR, &two, &f1), f2 = f_sub(&FR, &a, &f0);            /* two.mul(f_1).sub(f_0) */
            fe b = f_mul(&FR, &three, &f1), c = f_mul(&FR, &two, &f0), f3 = f_sub(&FR, &b, &c);  /* three.mul(f_1).sub(two.mul(f_0)) */
            fe f[4] = {f0, f1, f2, f3};
            for (int x = 0; x < 4; x++) v[x] = k == 0 ? f[x] : f_mul(&FR, &v[x], &f[x]);
        }
        for (int x = 0; x < 4; x++) e[x] = f_add(&FR, &e[x], &v[x]);
    }
    memcpy(evals, e, 128);
}
/* OutputSumcheckProver.computeRoundPolynomial's s(0..3) — src/zkvm/ram/output_check.zig:375-430 */
EXPORT void zo_output_check_round(const uint64_t *eq, const uint64_t *io, const uint64_t *vf, const uint64_t *vio, size_t current_size,
                                  uint64_t evals[16]) {
    fe s[4] = {f_zero(), f_zero(), f_zero(), f_zero()};
    const fe *E = (const fe *)eq, *IO = (const fe *)io, *VF = (const fe *)vf, *VIO = (const fe *)vio;
    for (size_t g = 0; g < current_size / 2; g++) {
        size_t i0 = 2 * g, i1 = 2 * g + 1;
        fe v0 = f_sub(&FR, &VF[i0], &VIO[i0]), v1 = f_sub(&FR, &VF[i1], &VIO[i1]);
        fe deq = f_sub(&FR, &E[i1], &E[i0]), dio = f_sub(&FR, &IO[i1], &IO[i0]), dv = f_sub(&FR, &v1, &v0);
        fe p0 = f_mul(&FR, &E[i0], &IO[i0]); p0 = f_mul(&FR, &p0, &v0);
        fe p1 = f_mul(&FR, &E[i1], &IO[i1]); p1 = f_mul(&FR, &p1, &v1);
        fe eq2 = f_add(&FR, &E[i0], &deq); eq2 = f_add(&FR, &eq2, &deq);
        fe io2 = f_add(&FR, &IO[i0], &dio); io2 = f_add(&FR, &io2, &dio);
        fe v2 = f_add(&FR, &v0, &dv); v2 = f_add(&FR, &v2, &dv);
        fe p2 = f_mul(&FR, &eq2, &io2); p2 = f_mul(&FR, &p2, &v2);
        fe eq3 = f_add(&FR, &eq2, &deq), io3 = f_add(&FR, &io2, &dio), v3 = f_add(&FR, &v2, &dv);
        fe p3 = f_mul(&FR, &eq3, &io3); p3 = f_mul(&FR, &p3, &v3);
        s[0] = f_add(&FR, &s[0], &p0); s[1] = f_add(&FR, &s[1], &p1); s[2] = f_add(&FR, &s[2], &p2); s[3] = f_add(&FR, &s[3], &p3);
    }
    memcpy(evals, s, 128);
}
/* OutputSumcheckProver.updateClaim — :482-499 with lagrangeC2 / lagrangeC3 (:523-548): c0 + c1 r + c2 r^2 + c3 r^3 */
EXPORT void zo_output_check_update_claim(const uint64_t evals[16], const uint64_t challenge[4], uint64_t out[4]) {
    const fe *e = (const fe *)evals;
    fe r = *(const fe *)challenge, r2 = f_mul(&FR, &r, &r), r3 = f_mul(&FR, &r2, &r);
    fe two = f_from_u64(&FR, 2), four = f_from_u64(&FR, 4), five = f_from_u64(&FR, 5), three = f_from_u64(&FR, 3), six = f_from_u64(&FR, 6);
    fe half, sixth, zero = f_zero();
    (void)f_inv(&FR, &two, &half); (void)f_inv(&FR, &six, &sixth);
    fe t, c2 = f_mul(&FR, &e[0], &two);
    t = f_mul(&FR, &e[1], &five); c2 = f_sub(&FR, &c2, &t);
    t = f_mul(&FR, &e[2], &four); c2 = f_add(&FR, &c2, &t);
    c2 = f_sub(&FR, &c2, &e[3]); c2 = f_mul(&FR, &c2, &half);
    fe c3 = f_sub(&FR, &zero, &e[0]);                        /* (-s(0) + 3 s(1) - 3 s(2) + s(3)) / 6 */
    t = f_mul(&FR, &e[1], &three); c3 = f_add(&FR, &c3, &t);
    t = f_mul(&FR, &e[2], &three); c3 = f_sub(&FR, &c3, &t);
    c3 = f_add(&FR, &c3, &e[3]); c3 = f_mul(&FR, &c3, &sixth);
    fe c1 = f_sub(&FR, &e[1], &e[0]); c1 = f_sub(&FR, &c1, &c2); c1 = f_sub(&FR, &c1, &c3);
    fe res = e[0];
    t = f_mul(&FR, &c1, &r); res = f_add(&FR, &res, &t);
    t = f_mul(&FR, &c2, &r2); res = f_add(&FR, &res, &t);
    t = f_mul(&FR, &c3, &r3); res = f_add(&FR, &res, &t);
    memcpy(out, &res, 32);
}
/* InstructionLookupsClaimReduction.computeRoundPolynomialCubic — src/zkvm/claim_reductions/instruction_lookups.zig:146-200 */
EXPORT void zo_instruction_lookups_round(const uint64_t *eq, const uint64_t *lookup_outputs, const uint64_t *left, const uint64_t *right,
                                         size_t current_len, const uint64_t gamma[4], const uint64_t current_claim[4], uint64_t evals[16]) {
    const fe *E = (const fe *)eq, *O = (const fe *)lookup_outputs, *L = (const fe *)left, *R = (const fe *)right;
    fe g = *(const fe *)gamma, g2 = f_mul(&FR, &g, &g), s0 = f_zero(), s2 = f_zero();
    for (size_t idx = 0; idx < current_len / 2; idx++) {
        size_t lo = 2 * idx, hi = 2 * idx + 1;
        fe a = f_mul(&FR, &g, &L[lo]), b = f_mul(&FR, &g2, &R[lo]), clo = f_add(&FR, &O[lo], &a); clo = f_add(&FR, &clo, &b);
        a = f_mul(&FR, &g, &L[hi]); b = f_mul(&FR, &g2, &R[hi]);
        fe chi = f_add(&FR, &O[hi], &a); chi = f_add(&FR, &chi, &b);
        fe prod0 = f_mul(&FR, &E[lo], &clo);
        fe eq2 = f_add(&FR, &E[hi], &E[hi]); eq2 = f_sub(&FR, &eq2, &E[lo]);
        fe c2 = f_add(&FR, &chi, &chi); c2 = f_sub(&FR, &c2, &clo);
        fe prod2 = f_mul(&FR, &eq2, &c2);
        s0 = f_add(&FR, &s0, &prod0); s2 = f_add(&FR, &s2, &prod2);
    }
    fe three = f_from_u64(&FR, 3), s1 = f_sub(&FR, (const fe *)current_claim, &s0);
    fe a = f_mul(&FR, &s1, &three), b = f_mul(&FR, &s2, &three), s3 = f_sub(&FR, &s0, &a); s3 = f_add(&FR, &s3, &b);
    fe out[4] = {s0, s1, s2, s3};
    memcpy(evals, out, 128);
}
/* ProductVirtualRemainderProver.computeRoundPolynomial's t0 / t_inf — src/zkvm/spartan/product_remainder.zig:281-330 */
EXPORT void zo_product_remainder_sums(const uint64_t *left, const uint64_t *right, size_t n, const uint64_t *e_out, size_t n_out,
                                      const uint64_t *e_in, size_t n_in, uint64_t t0[4], uint64_t t_inf[4]) {
    const fe *Lf = (const fe *)left, *Rt = (const fe *)right, *EO = (const fe *)e_out, *EI = (const fe *)e_in;
    size_t num_groups = n / 2;
    unsigned bits = 0;
    if (n_in > 1) while (((size_t)1 << bits) < n_in) bits++;
    fe t0s = f_zero(), tis = f_zero();
    for (size_t xo = 0; xo < n_out; xo++) {
        fe i0 = f_zero(), ii = f_zero();
        for (size_t xi = 0; xi < n_in; xi++) {
            size_t g = (xo << bits) | xi;
            if (g < num_groups) {
                fe p0 = f_mul(&FR, &Lf[2 * g], &Rt[2 * g]);
                fe dl = f_sub(&FR, &Lf[2 * g + 1], &Lf[2 * g]), dr = f_sub(&FR, &Rt[2 * g + 1], &Rt[2 * g]), sl = f_mul(&FR, &dl, &dr);
                fe a = f_mul(&FR, &p0, &EI[xi]), b = f_mul(&FR, &sl, &EI[xi]);
                i0 = f_add(&FR, &i0, &a); ii = f_add(&FR, &ii, &b);
            }
        }
        fe a = f_mul(&FR, &i0, &EO[xo]), b = f_mul(&FR, &ii, &EO[xo]);
        t0s = f_add(&FR, &t0s, &a); tis = f_add(&FR, &tis, &b);
    }
    memcpy(t0, &t0s, 32); memcpy(t_inf, &tis, 32);
}

/* R1CSInputEvaluator.computeClaimedInputs — src/zkvm/r1cs/evaluation.zig:55-122: rows = cycle_witnesses[t].values (k per cycle);
 * log_n = floor(log2(num_cycles)), padded_len = 2^log_n, effective_len = min(r_len, log_n); effective_len == 0 -> the first
 * witness's values; else out[i] = sum_{t < min(num_cycles, padded_len)} eq(r[0..effective_len))[t] * rows[t][i] with the eq table of
 * EqPolynomial.evals (r[0] <-> MSB). Returns 0, or -1 where the reference would index eq_evals out of bounds (effective_len < log_n). */
EXPORT int zo_r1cs_claimed_inputs(const uint64_t *rows, size_t num_cycles, size_t k, const uint64_t *r, size_t r_len, uint64_t *out) {
    fe *o = (fe *)out;
    const fe *W = (const fe *)rows;
    for (size_t i = 0; i < k; i++) o[i] = f_zero();
    if (num_cycles == 0) return 0;
    size_t log_n = 0;
    while (((size_t)2 << log_n) <= num_cycles) log_n++;
    size_t padded_len = (size_t)1 << log_n;
    size_t effective_len = r_len < log_n ? r_len : log_n;
    if (effective_len == 0) {
        for (size_t i = 0; i < k; i++) o[i] = W[i];
        return 0;
    }
    if (effective_len < log_n) return -1;
    size_t n_eq = (size_t)1 << effective_len;
    fe *eq = (fe *)malloc(n_eq * sizeof(fe));
    zo_fr_eq_table(r, effective_len, NULL, (uint64_t *)eq);
    size_t lim = num_cycles < padded_len ? num_cycles : padded_len;
    for (size_t t = 0; t < lim; t++) {
        if (f_is_zero(&eq[t])) continue;
        for (size_t i = 0; i < k; i++) {
            fe m = f_mul(&FR, &eq[t], &W[t * k + i]);
            o[i] = f_add(&FR, &o[i], &m);
        }
    }
    free(eq);
    return 0;
}

/* ---------------------------------------------------------- Stage-3 prover rounds (src/zkvm/spartan/stage3_prover.zig) */
/* ShiftSumcheckProver.computeRoundEvalsPhase1 — :1351-1392: four (P, Q) pairs, H(X) = sum_pairs sum_j P(X) Q(X) at X = 0, 1, 2 */
EXPORT void zo_shift_phase1_round(const uint64_t *const *P, const uint64_t *const *Q, size_t current_prefix_size, uint64_t evals[12]) {
    fe e[3] = {f_zero(), f_zero(), f_zero()};
    size_t half = current_prefix_size / 2;
    for (int k = 0; k < 4; k++) {
        const fe *p = (const fe *)P[k], *q = (const fe *)Q[k];
        for (size_t i = 0; i < half; i++) {
            fe p0 = p[2 * i], p1 = p[2 * i + 1], q0 = q[2 * i], q1 = q[2 * i + 1];
            fe p2 = f_add(&FR, &p1, &p1); p2 = f_sub(&FR, &p2, &p0);
            fe q2 = f_add(&FR, &q1, &q1); q2 = f_sub(&FR, &q2, &q0);
            fe a = f_mul(&FR, &p0, &q0), b = f_mul(&FR, &p1, &q1), c = f_mul(&FR, &p2, &q2);
            e[0] = f_add(&FR, &e[0], &a); e[1] = f_add(&FR, &e[1], &b); e[2] = f_add(&FR, &e[2], &c);
        }
    }
    memcpy(evals, e, 96);
}
static fe lin2(const fe *t, size_t j) {  /* X = 2 extrapolation of the pair (2j, 2j+1): f1 + f1 - f0 */
    fe v = f_add(&FR, &t[2 * j + 1], &t[2 * j + 1]);
    return f_sub(&FR, &v, &t[2 * j]);
}
/* ShiftSumcheckProver.computeRoundEvalsPhase2 — :1399-1455: tables eq_outer, eq_prod, upc, pc, virt, first, noop; gamma_powers[0..4];
 * -> [p(0), previous_claim - p(0), p(2)] */
EXPORT void zo_shift_phase2_round(const uint64_t *const *tabs, size_t n, const uint64_t *gamma_powers, const uint64_t previous_claim[4],
                                  uint64_t evals[12]) {
    const fe *eo = (const fe *)tabs[0], *ep = (const fe *)tabs[1], *upc = (const fe *)tabs[2], *pc = (const fe *)tabs[3],
             *virt = (const fe *)tabs[4], *first = (const fe *)tabs[5], *noop = (const fe *)tabs[6];
    const fe *g = (const fe *)gamma_powers;
    fe one = f_one(&FR), e0 = f_zero(), e2 = f_zero();
    for (size_t j = 0; j < n / 2; j++) {
        for (int x = 0; x < 2; x++) {  /* X = 0, then X = 2 */
            fe veo = x ? lin2(eo, j) : eo[2 * j], vep = x ? lin2(ep, j) : ep[2 * j], vu = x ? lin2(upc, j) : upc[2 * j];
            fe vp = x ? lin2(pc, j) : pc[2 * j], vv = x ? lin2(virt, j) : virt[2 * j], vf = x ? lin2(first, j) : first[2 * j];
            fe vn = x ? lin2(noop, j) : noop[2 * j];
            fe a = f_mul(&FR, &g[1], &vp), b = f_mul(&FR, &g[2], &vv), c = f_mul(&FR, &g[3], &vf);
            fe val = f_add(&FR, &vu, &a); val = f_add(&FR, &val, &b); val = f_add(&FR, &val, &c);
            fe term1 = f_mul(&FR, &veo, &val);
            fe omn = f_sub(&FR, &one, &vn), term2 = f_mul(&FR, &g[4], &omn); term2 = f_mul(&FR, &term2, &vep);
            fe f = f_add(&FR, &term1, &term2);
            if (x) e2 = f_add(&FR, &e2, &f); else e0 = f_add(&FR, &e0, &f);
        }
    }
    fe p1 = f_sub(&FR, (const fe *)previous_claim, &e0);
    fe out[3] = {e0, p1, e2};
    memcpy(evals, out, 96);
}
/* InstructionInputProver.computeRoundEvals — :2029-2100: tables left_is_rs1, rs1_value, left_is_pc, unexpanded_pc, right_is_rs2, rs2_value,
 * right_is_imm, imm, eq_outer, eq_product; -> [p(0), previous_claim - p(0), p(2), p(3)] */
EXPORT void zo_instruction_input_round(const uint64_t *const *tabs, size_t n, const uint64_t gamma[4], const uint64_t previous_claim[4],
                                       uint64_t evals[16]) {
    const fe *T[10];
    for (int k = 0; k < 10; k++) T[k] = (const fe *)tabs[k];
    fe g = *(const fe *)gamma, g2 = f_mul(&FR, &g, &g), e[3] = {f_zero(), f_zero(), f_zero()};
    for (size_t j = 0; j < n / 2; j++) {
        fe v[10][3];  /* values at X = 0, 2, 3: f_2 = f_1 + f_1 - f_0, f_3 = f_2 + f_1 - f_0 */
        for (int k = 0; k < 10; k++) {
            fe f0 = T[k][2 * j], f1 = T[k][2 * j + 1];
            fe f2 = f_add(&FR, &f1, &f1); f2 = f_sub(&FR, &f2, &f0);
            fe f3 = f_add(&FR, &f2, &f1); f3 = f_sub(&FR, &f3, &f0);
            v[k][0] = f0; v[k][1] = f2; v[k][2] = f3;
        }
        for (int x = 0; x < 3; x++) {
            fe a = f_mul(&FR, &v[0][x], &v[1][x]), b = f_mul(&FR, &v[2][x], &v[3][x]), left = f_add(&FR, &a, &b);
            a = f_mul(&FR, &v[4][x], &v[5][x]); b = f_mul(&FR, &v[6][x], &v[7][x]);
            fe right = f_add(&FR, &a, &b);
            fe w = f_mul(&FR, &g2, &v[9][x]), eqw = f_add(&FR, &v[8][x], &w);
            fe gl = f_mul(&FR, &g, &left), sum = f_add(&FR, &right, &gl), f = f_mul(&FR, &eqw, &sum);
            e[x] = f_add(&FR, &e[x], &f);
        }
    }
    fe p1 = f_sub(&FR, (const fe *)previous_claim, &e[0]);
    fe out[4] = {e[0], p1, e[1], e[2]};
    memcpy(evals, out, 128);
}
/* RegistersClaimReductionProver.computeRoundEvalsPhase1 (:2334-2354: P * Q) and Phase2 (:2356-2389: eq * (rd + gamma rs1 + gamma^2 rs2));
 * phase 1: tabs = {P, Q}; phase 2: tabs = {eq, rd_write_value, rs1_value, rs2_value}; -> [p(0), previous_claim - p(0), p(2)] */
EXPORT void zo_registers_cr_round(int phase2, const uint64_t *const *tabs, size_t n, const uint64_t gamma[4], const uint64_t previous_claim[4],
                                  uint64_t evals[12]) {
    fe e0 = f_zero(), e2 = f_zero();
    if (!phase2) {
        const fe *P = (const fe *)tabs[0], *Q = (const fe *)tabs[1];
        for (size_t i = 0; i < n / 2; i++) {
            fe p2 = lin2(P, i), q2 = lin2(Q, i), a = f_mul(&FR, &P[2 * i], &Q[2 * i]), b = f_mul(&FR, &p2, &q2);
            e0 = f_add(&FR, &e0, &a); e2 = f_add(&FR, &e2, &b);
        }
    } else {
        const fe *eq = (const fe *)tabs[0], *rd = (const fe *)tabs[1], *rs1 = (const fe *)tabs[2], *rs2 = (const fe *)tabs[3];
        fe g = *(const fe *)gamma, g2 = f_mul(&FR, &g, &g);
        for (size_t j = 0; j < n / 2; j++) {
            fe eq2 = lin2(eq, j), rd2 = lin2(rd, j), a2 = lin2(rs1, j), b2 = lin2(rs2, j);
            fe x = f_mul(&FR, &g, &rs1[2 * j]), y = f_mul(&FR, &g2, &rs2[2 * j]), v0 = f_add(&FR, &rd[2 * j], &x); v0 = f_add(&FR, &v0, &y);
            x = f_mul(&FR, &g, &a2); y = f_mul(&FR, &g2, &b2);
            fe v2 = f_add(&FR, &rd2, &x); v2 = f_add(&FR, &v2, &y);
            fe a = f_mul(&FR, &eq[2 * j], &v0), b = f_mul(&FR, &eq2, &v2);
            e0 = f_add(&FR, &e0, &a); e2 = f_add(&FR, &e2, &b);
        }
    }
    fe p1 = f_sub(&FR, (const fe *)previous_claim, &e0);
    fe out[3] = {e0, p1, e2};
    memcpy(evals, out, 96);
}

/* EqPlusOnePolynomial.mle — src/poly/mod.zig:407-435 (x, y big-endian: index 0 is the MSB) */
EXPORT void zo_eq_plus_one_mle(const uint64_t *x, const uint64_t *y, size_t l, uint64_t out[4]) {
    const fe *X = (const fe *)x, *Y = (const fe *)y;
    fe one = f_one(&FR), result = f_zero();
    for (size_t k = 0; k < l; k++) {
        fe lower = one;
        for (size_t i = 0; i < k; i++) {
            size_t idx = l - 1 - i;
            fe omy = f_sub(&FR, &one, &Y[idx]), t = f_mul(&FR, &X[idx], &omy);
            lower = f_mul(&FR, &lower, &t);
        }
        size_t kth_idx = l - 1 - k;
        fe omx = f_sub(&FR, &one, &X[kth_idx]), kth = f_mul(&FR, &omx, &Y[kth_idx]);
        fe higher = one;
        for (size_t i = k + 1; i < l; i++) {
            size_t idx = l - 1 - i;
            fe xy = f_mul(&FR, &X[idx], &Y[idx]), a = f_sub(&FR, &one, &X[idx]), b = f_sub(&FR, &one, &Y[idx]), ab = f_mul(&FR, &a, &b);
            fe s = f_add(&FR, &xy, &ab);
            higher = f_mul(&FR, &higher, &s);
        }
        fe t = f_mul(&FR, &lower, &kth); t = f_mul(&FR, &t, &higher);
        result = f_add(&FR, &result, &t);
    }
    memcpy(out, &result, 32);
}
/* computeEqPlusOneEvals — src/poly/mod.zig:530-548: out[j] = mle(r, bits of j as field elements, MSB first), the reference's way */
EXPORT void zo_eq_plus_one_table(const uint64_t *r, size_t n, uint64_t *out) {
    size_t size = (size_t)1 << n;
    fe *bits = (fe *)malloc((n ? n : 1) * sizeof(fe));
    for (size_t j = 0; j < size; j++) {
        for (size_t k = 0; k < n; k++) bits[k] = ((j >> (n - 1 - k)) & 1) ? f_one(&FR) : f_zero();
        zo_eq_plus_one_mle(r, (const uint64_t *)bits, n, out + 4 * j);
    }
    free(bits);
}


/* LtPolynomial.evaluateAtIndex over the cube — src/zkvm/ram/val_evaluation.zig:309-330, the reference's double loop (index bit i <-> r[i]) */
EXPORT void zo_lt_table(const uint64_t *r, size_t v, uint64_t *out) {
    const fe *R = (const fe *)r;
    fe one = f_one(&FR);
    for (size_t j = 0; j < ((size_t)1 << v); j++) {
        fe result = f_zero();
        for (size_t i = 0; i < v; i++) {
            if ((j >> i) & 1) continue;
            fe contrib = R[i];
            for (size_t k = i + 1; k < v; k++) {
                fe omr = f_sub(&FR, &one, &R[k]);
                contrib = f_mul(&FR, &contrib, ((j >> k) & 1) ? &R[k] : &omr);
            }
            result = f_add(&FR, &result, &contrib);
        }
        memcpy(out + 4 * j, &result, 32);
    }
}
/* the x_hi / x_lo double loop of the Stage-3 Q tables and Dory's vector-matrix product — src/zkvm/spartan/stage3_prover.zig:1066-1112,
 * src/poly/commitment/dory.zig:622-642: out[k][c] = sum_r weights[k][r] * table[r * cols + c] */
EXPORT void zo_weighted_colsum(const uint64_t *table, size_t rows, size_t cols, const uint64_t *weights, size_t m, uint64_t *out) {
    const fe *T = (const fe *)table, *W = (const fe *)weights;
    fe *O = (fe *)out;
    for (size_t i = 0; i < m * cols; i++) O[i] = f_zero();
    for (size_t r = 0; r < rows; r++)
        for (size_t c = 0; c < cols; c++)
            for (size_t k = 0; k < m; k++) {
                fe p = f_mul(&FR, &T[r * cols + c], &W[k * rows + r]);
                O[k * cols + c] = f_add(&FR, &O[k * cols + c], &p);
            }
}
