/*
 * oracle/zk_oracle.c — CPU restatement of the snarkjs prover hot path.   *** TEST INFRASTRUCTURE ONLY ***
 *
 * This file is the parity ORACLE for the MI355X backend. It is never linked into, imported by, or called from
 * the product path (snarkjs_amd/, include/zkmi.h). Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it.
 *
 * What it restates: the arithmetic of ffjavascript 0.3.1 + wasmcurves 0.2.2 (third-party
 * dependencies of the reference, pinned in /root/reference/package.json:65 and package-lock.json; present in the
 * reference tree only as the minified bundle build/snarkjs.min.js — cited below as `min.js:1@<column>`) and the
 * Groth16 prover driver src/groth16_prove.js of snarkjs 0.7.6.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function here against golden vectors produced by the
 * reference itself, run in the build container through oracle/ref_shim.js + oracle/gen_golden.js
 * (tests/golden/: NTT / batch ops / MSM outputs on both curves, and a fully seeded Groth16 proof whose JSON hash
 * equals SURVEY.md Appendix C.3).
 *
 * Conventions restated from the reference:
 *   - field elements: little-endian, n8 bytes; "M" = Montgomery form x·2^(8·n8) mod p, fully reduced.
 *   - affine points: (x,y) M; the point at infinity is all-zero bytes. Jacobian: (X,Y,Z) M; zero has Z = 0.
 *   - G2 over Fq2 = Fq[u]/(u^2+1), element = (c0,c1).
 *   - Fr.w[i]: w[s] = nqr^((r-1)/2^s), w[i] = w[i+1]^2; nqr = smallest quadratic non-residue (min.js:1@185893).
 *
 * Build: make -C oracle   →  oracle/libzkoracle.so   (plain C11, gcc, no dependencies)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;
#define MAXL 6              /* 6 x 64-bit limbs = BLS12-381 Fq */
#define MAXE (2 * MAXL)     /* Fq2 element */

/* ------------------------------------------------------------------------------------------------------------ */
/* Prime field, Montgomery representation with R = 2^(64 n)                                                      */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int n;            /* 64-bit limbs */
    u64 p[MAXL];
    u64 np;           /* -p^{-1} mod 2^64 */
    u64 one[MAXL];    /* R mod p */
    u64 r2[MAXL];     /* R^2 mod p */
} fld;

static int  bn_cmp(const u64 *a, const u64 *b, int n) { for (int i = n - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; } return 0; }
static int  bn_is_zero(const u64 *a, int n) { u64 o = 0; for (int i = 0; i < n; i++) o |= a[i]; return o == 0; }
static u64  bn_add(u64 *r, const u64 *a, const u64 *b, int n) { u128 c = 0; for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; r[i] = (u64)c; c >>= 64; } return (u64)c; }
static u64  bn_sub(u64 *r, const u64 *a, const u64 *b, int n) { u64 bw = 0; for (int i = 0; i < n; i++) { u128 d = (u128)a[i] - b[i] - bw; r[i] = (u64)d; bw = (u64)(d >> 64) & 1; } return bw; }

static void fe_add(const fld *F, u64 *r, const u64 *a, const u64 *b) {
    u64 t[MAXL], c = bn_add(t, a, b, F->n);
    if (c || bn_cmp(t, F->p, F->n) >= 0) bn_sub(t, t, F->p, F->n);
    memcpy(r, t, 8 * F->n);
}
static void fe_sub(const fld *F, u64 *r, const u64 *a, const u64 *b) {
    u64 t[MAXL];
    if (bn_sub(t, a, b, F->n)) bn_add(t, t, F->p, F->n);
    memcpy(r, t, 8 * F->n);
}
static void fe_neg(const fld *F, u64 *r, const u64 *a) {
    if (bn_is_zero(a, F->n)) memset(r, 0, 8 * F->n); else bn_sub(r, F->p, a, F->n);
}
/* Montgomery product a·b·R^{-1} mod p: schoolbook product then word-by-word REDC (the reference's f1m_mul is a
 * product-scanning Montgomery on 32-bit limbs, min.js:1@36990; same function of (a,b)). */
static void fe_mul(const fld *F, u64 *r, const u64 *a, const u64 *b) {
    const int n = F->n;
    u64 t[2 * MAXL + 1];
    memset(t, 0, sizeof t);
    for (int i = 0; i < n; i++) {
        u128 c = 0;
        for (int j = 0; j < n; j++) { c += (u128)a[i] * b[j] + t[i + j]; t[i + j] = (u64)c; c >>= 64; }
        t[i + n] = (u64)c;
    }
    for (int i = 0; i < n; i++) {
        u64 m = t[i] * F->np;
        u128 c = 0;
        for (int j = 0; j < n; j++) { c += (u128)m * F->p[j] + t[i + j]; t[i + j] = (u64)c; c >>= 64; }
        for (int k = i + n; c && k <= 2 * n; k++) { c += t[k]; t[k] = (u64)c; c >>= 64; }
    }
    if (t[2 * n] || bn_cmp(t + n, F->p, n) >= 0) bn_sub(t + n, t + n, F->p, n);
    memcpy(r, t + n, 8 * n);
}
static void fe_sqr(const fld *F, u64 *r, const u64 *a) { fe_mul(F, r, a, a); }
static void fe_to_mont(const fld *F, u64 *r, const u64 *a) { fe_mul(F, r, a, F->r2); }
static void fe_from_mont(const fld *F, u64 *r, const u64 *a) { u64 o[MAXL] = {1}; fe_mul(F, r, a, o); }
/* r = a^e, e a plain little-endian integer of ne limbs; a, r Montgomery */
static void fe_pow(const fld *F, u64 *r, const u64 *a, const u64 *e, int ne) {
    u64 acc[MAXL], base[MAXL];
    memcpy(acc, F->one, 8 * F->n); memcpy(base, a, 8 * F->n);
    for (int i = 0; i < 64 * ne; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fe_mul(F, acc, acc, base);
        fe_sqr(F, base, base);
    }
    memcpy(r, acc, 8 * F->n);
}
static void fe_inv(const fld *F, u64 *r, const u64 *a) {          /* Fermat; 0 -> 0 */
    u64 e[MAXL], two[MAXL] = {2};
    bn_sub(e, F->p, two, F->n);
    fe_pow(F, r, a, e, F->n);
}
static void fld_init(fld *F, int n, const u64 *p) {
    memset(F, 0, sizeof *F);
    F->n = n; memcpy(F->p, p, 8 * n);
    u64 inv = 1;                                   /* Newton: inv = p^{-1} mod 2^64 */
    for (int i = 0; i < 6; i++) inv *= 2 - p[0] * inv;
    F->np = (u64)0 - inv;
    u64 t[MAXL] = {1};                              /* t = 2^(64n) mod p, then 2^(128n) mod p by repeated doubling */
    for (int i = 0; i < 128 * n; i++) {
        u64 c = bn_add(t, t, t, n);
        if (c || bn_cmp(t, p, n) >= 0) bn_sub(t, t, p, n);
        if (i == 64 * n - 1) memcpy(F->one, t, 8 * n);
    }
    memcpy(F->r2, t, 8 * n);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Curves                                                                                                        */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    fld Fq, Fr;
    int s;                       /* 2-adicity of r-1 */
    u64 w[33][MAXL];             /* Fr.w[i], Montgomery */
    u64 wi[33][MAXL];            /* inverses */
    u64 g1[2 * MAXL];            /* generator, affine M */
    u64 g2[4 * MAXL];
    int ready;
} curve_t;

static curve_t CURVES[2];

static const u64 BN254_Q[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 BN254_R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 BLS_Q[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 BLS_R[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};

/* generators in normal form (standard constants; checked against tests/golden/{curve}_kernel_vectors.json G1_g/G2_g) */
static const u64 BN254_G2[4][4] = {
    {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL},
    {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL},
    {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL},
    {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL}};
static const u64 BLS_G1[2][6] = {
    {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL},
    {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL}};
static const u64 BLS_G2[4][6] = {
    {0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL},
    {0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL},
    {0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL},
    {0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL}};

/* Fr.w[] exactly as ffjavascript builds it (min.js:1@185893): nqr = smallest x >= 2 with x^((r-1)/2) = -1;
 * s = 2-adicity; w[s] = nqr^((r-1) >> s); w[i] = w[i+1]^2. */
static void curve_init_roots(curve_t *C) {
    const fld *F = &C->Fr;
    u64 e[MAXL], rm1[MAXL], one[MAXL] = {1}, half[MAXL], negone[MAXL];
    bn_sub(rm1, F->p, one, F->n);
    int s = 0; memcpy(e, rm1, 8 * F->n);
    while (!(e[0] & 1)) { for (int i = 0; i < F->n; i++) e[i] = (e[i] >> 1) | (i + 1 < F->n ? e[i + 1] << 63 : 0); s++; }
    C->s = s;
    memcpy(half, rm1, 8 * F->n);
    for (int i = 0; i < F->n; i++) half[i] = (half[i] >> 1) | (i + 1 < F->n ? half[i + 1] << 63 : 0);
    fe_neg(F, negone, F->one);
    u64 nqr[MAXL], t[MAXL];
    for (u64 x = 2;; x++) {
        u64 xn[MAXL] = {x}; fe_to_mont(F, nqr, xn);
        fe_pow(F, t, nqr, half, F->n);
        if (bn_cmp(t, negone, F->n) == 0) break;
    }
    fe_pow(F, C->w[s], nqr, e, F->n);
    for (int i = s - 1; i >= 0; i--) fe_sqr(F, C->w[i], C->w[i + 1]);
    for (int i = 0; i <= s; i++) fe_inv(F, C->wi[i], C->w[i]);
}
static curve_t *get_curve(int id) {
    curve_t *C = &CURVES[id];
    if (C->ready) return C;
    if (id == 0) {
        fld_init(&C->Fq, 4, BN254_Q); fld_init(&C->Fr, 4, BN254_R);
        u64 gx[MAXL] = {1}, gy[MAXL] = {2};
        fe_to_mont(&C->Fq, C->g1, gx); fe_to_mont(&C->Fq, C->g1 + 4, gy);
        for (int k = 0; k < 4; k++) fe_to_mont(&C->Fq, C->g2 + 4 * k, BN254_G2[k]);
    } else {
        fld_init(&C->Fq, 6, BLS_Q); fld_init(&C->Fr, 4, BLS_R);
        for (int k = 0; k < 2; k++) fe_to_mont(&C->Fq, C->g1 + 6 * k, BLS_G1[k]);
        for (int k = 0; k < 4; k++) fe_to_mont(&C->Fq, C->g2 + 6 * k, BLS_G2[k]);
    }
    curve_init_roots(C);
    C->ready = 1;
    return C;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Extension-generic element ops: deg 1 (Fq, G1) or deg 2 (Fq2, G2). Elements are u64[deg*n].                    */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct { const fld *F; int deg; int L; /* limbs per element */ } ext;

static void e_add(const ext *E, u64 *r, const u64 *a, const u64 *b) { for (int k = 0; k < E->deg; k++) fe_add(E->F, r + k * E->F->n, a + k * E->F->n, b + k * E->F->n); }
static void e_sub(const ext *E, u64 *r, const u64 *a, const u64 *b) { for (int k = 0; k < E->deg; k++) fe_sub(E->F, r + k * E->F->n, a + k * E->F->n, b + k * E->F->n); }
static void e_neg(const ext *E, u64 *r, const u64 *a) { for (int k = 0; k < E->deg; k++) fe_neg(E->F, r + k * E->F->n, a + k * E->F->n); }
static void e_mul(const ext *E, u64 *r, const u64 *a, const u64 *b) {
    const fld *F = E->F; const int n = F->n;
    if (E->deg == 1) { fe_mul(F, r, a, b); return; }
    u64 t0[MAXL], t1[MAXL], t2[MAXL], t3[MAXL];      /* (a0+a1 u)(b0+b1 u), u^2 = -1 */
    fe_mul(F, t0, a, b); fe_mul(F, t1, a + n, b + n); fe_mul(F, t2, a, b + n); fe_mul(F, t3, a + n, b);
    fe_sub(F, r, t0, t1); fe_add(F, r + n, t2, t3);
}
static void e_sqr(const ext *E, u64 *r, const u64 *a) { u64 t[MAXE]; memcpy(t, a, 8 * E->L); e_mul(E, r, t, t); }
static int  e_is_zero(const ext *E, const u64 *a) { return bn_is_zero(a, E->L); }
static int  e_eq(const ext *E, const u64 *a, const u64 *b) { return memcmp(a, b, 8 * E->L) == 0; }
static void e_inv(const ext *E, u64 *r, const u64 *a) {
    const fld *F = E->F; const int n = F->n;
    if (E->deg == 1) { fe_inv(F, r, a); return; }
    u64 t0[MAXL], t1[MAXL], d[MAXL];                  /* 1/(a0+a1u) = (a0 - a1 u)/(a0^2+a1^2) */
    fe_sqr(F, t0, a); fe_sqr(F, t1, a + n); fe_add(F, d, t0, t1); fe_inv(F, d, d);
    fe_mul(F, r, a, d); fe_mul(F, t0, a + n, d); fe_neg(F, r + n, t0);
}
static void e_one(const ext *E, u64 *r) { memset(r, 0, 8 * E->L); memcpy(r, E->F->one, 8 * E->F->n); }
static void e_dbl(const ext *E, u64 *r, const u64 *a) { e_add(E, r, a, a); }

/* Jacobian points: u64[3L] = X|Y|Z. Affine: u64[2L]. Curves have a = 0. */
static void pt_zero(const ext *E, u64 *P) { memset(P, 0, 8 * 3 * E->L); }
static int  pt_is_zero(const ext *E, const u64 *P) { return e_is_zero(E, P + 2 * E->L); }
static void pt_double(const ext *E, u64 *R, const u64 *P) {
    const int L = E->L;
    if (pt_is_zero(E, P)) { pt_zero(E, R); return; }
    u64 A[MAXE], B[MAXE], C[MAXE], D[MAXE], Ee[MAXE], Ff[MAXE], t[MAXE], X3[MAXE], Y3[MAXE], Z3[MAXE];
    e_sqr(E, A, P); e_sqr(E, B, P + L); e_sqr(E, C, B);
    e_add(E, t, P, B); e_sqr(E, t, t); e_sub(E, t, t, A); e_sub(E, t, t, C); e_dbl(E, D, t);
    e_dbl(E, Ee, A); e_add(E, Ee, Ee, A); e_sqr(E, Ff, Ee);
    e_dbl(E, t, D); e_sub(E, X3, Ff, t);
    e_mul(E, Z3, P + L, P + 2 * L); e_dbl(E, Z3, Z3);
    e_sub(E, t, D, X3); e_mul(E, Y3, Ee, t); e_dbl(E, t, C); e_dbl(E, t, t); e_dbl(E, t, t); e_sub(E, Y3, Y3, t);
    memcpy(R, X3, 8 * L); memcpy(R + L, Y3, 8 * L); memcpy(R + 2 * L, Z3, 8 * L);
}
/* general addition with all special cases (P = Q -> double, P = -Q -> zero, zero operands) */
static void pt_add(const ext *E, u64 *R, const u64 *P, const u64 *Q) {
    const int L = E->L;
    if (pt_is_zero(E, P)) { memmove(R, Q, 8 * 3 * L); return; }
    if (pt_is_zero(E, Q)) { memmove(R, P, 8 * 3 * L); return; }
    u64 Z1Z1[MAXE], Z2Z2[MAXE], U1[MAXE], U2[MAXE], S1[MAXE], S2[MAXE], H[MAXE], I[MAXE], J[MAXE], r[MAXE], V[MAXE], t[MAXE], X3[MAXE], Y3[MAXE], Z3[MAXE];
    e_sqr(E, Z1Z1, P + 2 * L); e_sqr(E, Z2Z2, Q + 2 * L);
    e_mul(E, U1, P, Z2Z2); e_mul(E, U2, Q, Z1Z1);
    e_mul(E, S1, P + L, Q + 2 * L); e_mul(E, S1, S1, Z2Z2);
    e_mul(E, S2, Q + L, P + 2 * L); e_mul(E, S2, S2, Z1Z1);
    if (e_eq(E, U1, U2)) {
        if (e_eq(E, S1, S2)) { pt_double(E, R, P); return; }
        pt_zero(E, R); return;
    }
    e_sub(E, H, U2, U1); e_dbl(E, I, H); e_sqr(E, I, I); e_mul(E, J, H, I);
    e_sub(E, r, S2, S1); e_dbl(E, r, r); e_mul(E, V, U1, I);
    e_sqr(E, X3, r); e_sub(E, X3, X3, J); e_sub(E, X3, X3, V); e_sub(E, X3, X3, V);
    e_sub(E, t, V, X3); e_mul(E, Y3, r, t); e_mul(E, t, S1, J); e_dbl(E, t, t); e_sub(E, Y3, Y3, t);
    e_add(E, Z3, P + 2 * L, Q + 2 * L); e_sqr(E, Z3, Z3); e_sub(E, Z3, Z3, Z1Z1); e_sub(E, Z3, Z3, Z2Z2); e_mul(E, Z3, Z3, H);
    memcpy(R, X3, 8 * L); memcpy(R + L, Y3, 8 * L); memcpy(R + 2 * L, Z3, 8 * L);
}
static int  aff_is_zero(const ext *E, const u64 *A) { return bn_is_zero(A, 2 * E->L); }
static void pt_from_affine(const ext *E, u64 *P, const u64 *A) {
    if (aff_is_zero(E, A)) { pt_zero(E, P); return; }
    memcpy(P, A, 8 * 2 * E->L); e_one(E, P + 2 * E->L);
}
static void pt_add_affine(const ext *E, u64 *R, const u64 *P, const u64 *A) {
    u64 Q[3 * MAXE]; pt_from_affine(E, Q, A); pt_add(E, R, P, Q);
}
static void pt_neg(const ext *E, u64 *R, const u64 *P) { memmove(R, P, 8 * 3 * E->L); e_neg(E, R + E->L, P + E->L); }
static void pt_to_affine(const ext *E, u64 *A, const u64 *P) {
    const int L = E->L;
    if (pt_is_zero(E, P)) { memset(A, 0, 8 * 2 * L); return; }
    u64 zi[MAXE], zi2[MAXE], zi3[MAXE];
    e_inv(E, zi, P + 2 * L); e_sqr(E, zi2, zi); e_mul(E, zi3, zi2, zi);
    e_mul(E, A, P, zi2); e_mul(E, A + L, P + L, zi3);
}
/* R = k·P, k a plain little-endian integer of nb bytes (double-and-add, MSB first) */
static void pt_mul_bytes(const ext *E, u64 *R, const u64 *P, const uint8_t *k, int nb) {
    u64 acc[3 * MAXE]; pt_zero(E, acc);
    for (int i = 8 * nb - 1; i >= 0; i--) {
        pt_double(E, acc, acc);
        if ((k[i / 8] >> (i % 8)) & 1) pt_add(E, acc, acc, P);
    }
    memcpy(R, acc, 8 * 3 * E->L);
}
static ext make_ext(const curve_t *C, int group) { ext E; E.F = &C->Fq; E.deg = group; E.L = group * C->Fq.n; return E; }

/* ------------------------------------------------------------------------------------------------------------ */
/* Exported API (ctypes). curve: 0 = bn128 (BN254), 1 = bls12381. group: 1 = G1, 2 = G2. Returns 0 on success.    */
/* ------------------------------------------------------------------------------------------------------------ */
int orc_n8q(int curve) { return 8 * get_curve(curve)->Fq.n; }
int orc_n8r(int curve) { return 8 * get_curve(curve)->Fr.n; }
int orc_two_adicity(int curve) { return get_curve(curve)->s; }
void orc_fr_w(int curve, int i, uint8_t *out) { curve_t *C = get_curve(curve); memcpy(out, C->w[i], 8 * C->Fr.n); }
void orc_fr_one(int curve, uint8_t *out) { curve_t *C = get_curve(curve); memcpy(out, C->Fr.one, 8 * C->Fr.n); }
void orc_fq_one(int curve, uint8_t *out) { curve_t *C = get_curve(curve); memcpy(out, C->Fq.one, 8 * C->Fq.n); }
void orc_generator(int curve, int group, uint8_t *out) { curve_t *C = get_curve(curve); memcpy(out, group == 1 ? C->g1 : C->g2, 8 * 2 * group * C->Fq.n); }

/* Fr element ops on 32-byte M values (for tests of the host glue) */
void orc_fr_mul(int curve, const uint8_t *a, const uint8_t *b, uint8_t *r) { curve_t *C = get_curve(curve); u64 x[MAXL], y[MAXL], z[MAXL]; memcpy(x, a, 32); memcpy(y, b, 32); fe_mul(&C->Fr, z, x, y); memcpy(r, z, 32); }
void orc_fr_from_u64(int curve, u64 v, uint8_t *r) { curve_t *C = get_curve(curve); u64 x[MAXL] = {v}, z[MAXL]; fe_to_mont(&C->Fr, z, x); memcpy(r, z, 32); }

/* Fr.batchToMontgomery / batchFromMontgomery (helper before min.js:1@185893): per element ×R / ×R^{-1}. */
int orc_fr_batch_to_mont(int curve, const uint8_t *in, uint8_t *out, size_t n) {
    curve_t *C = get_curve(curve); u64 x[MAXL], y[MAXL];
    for (size_t i = 0; i < n; i++) { memcpy(x, in + 32 * i, 32); fe_to_mont(&C->Fr, y, x); memcpy(out + 32 * i, y, 32); }
    return 0;
}
int orc_fr_batch_from_mont(int curve, const uint8_t *in, uint8_t *out, size_t n) {
    curve_t *C = get_curve(curve); u64 x[MAXL], y[MAXL];
    for (size_t i = 0; i < n; i++) { memcpy(x, in + 32 * i, 32); fe_from_mont(&C->Fr, y, x); memcpy(out + 32 * i, y, 32); }
    return 0;
}
/* Fr.batchInverse (min.js:1@188677): element-wise inverse, 0 -> 0. (The reference uses Montgomery's trick per worker
 * slice with zeros skipped; the function computed is the element-wise inverse.) */
int orc_fr_batch_inverse(int curve, const uint8_t *in, uint8_t *out, size_t n) {
    curve_t *C = get_curve(curve);
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) { u64 x[MAXL], y[MAXL]; memcpy(x, in + 32 * i, 32); fe_inv(&C->Fr, y, x); memcpy(out + 32 * i, y, 32); }
    return 0;
}
/* Fr.batchApplyKey(buf, first, inc) (min.js:1@211529, kernel frm_batchApplyKey @128060): out[i] = in[i]·first·inc^i.
 * first/inc are M elements. */
int orc_fr_batch_apply_key(int curve, const uint8_t *in, uint8_t *out, size_t n, const uint8_t *first, const uint8_t *inc) {
    curve_t *C = get_curve(curve); u64 f0[MAXL], k[MAXL];
    memcpy(f0, first, 32); memcpy(k, inc, 32);
    const size_t CH = 4096;
    #pragma omp parallel for schedule(static)
    for (size_t c0 = 0; c0 < n; c0 += CH) {
        u64 x[MAXL], y[MAXL], t[MAXL], e[1] = {(u64)c0};
        fe_pow(&C->Fr, t, k, e, 1); fe_mul(&C->Fr, t, t, f0);                 /* first * inc^c0 */
        size_t hi = c0 + CH < n ? c0 + CH : n;
        for (size_t i = c0; i < hi; i++) { memcpy(x, in + 32 * i, 32); fe_mul(&C->Fr, y, x, t); memcpy(out + 32 * i, y, 32); fe_mul(&C->Fr, t, t, k); }
    }
    return 0;
}
/* Fr.fft / Fr.ifft (min.js:1@215859 driver; kernels frm_fftMix/_fftJoin/_fftFinal @103755):
 *   X[k] = sum_j x[j]·w^(jk), w = Fr.w[log2 n], natural order in and out; ifft is the exact inverse incl. 1/n.
 * Restated as bit-reversal + iterative radix-2 decimation-in-time (what fftMix/fftJoin compute block-wise). */
int orc_fr_ntt(int curve, const uint8_t *in, uint8_t *out, unsigned log_n, int inverse) {
    curve_t *C = get_curve(curve); const fld *F = &C->Fr;
    if ((int)log_n > C->s) return -1;
    size_t n = (size_t)1 << log_n;
    u64 (*a)[4] = malloc(n * 32);
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        size_t r = 0; for (unsigned b = 0; b < log_n; b++) r |= ((i >> b) & 1) << (log_n - 1 - b);
        memcpy(a[r], in + 32 * i, 32);
    }
    /* one table of the n/2 powers of w = Fr.w[log_n]: stage st uses every (n/2h)-th entry (w_st = w^(n/2h)) */
    size_t half = n > 1 ? n / 2 : 1;
    u64 (*tw)[4] = malloc(half * 32);
    {
        const u64 *wn = inverse ? C->wi[log_n] : C->w[log_n];
        const size_t CH = 4096;
        #pragma omp parallel for schedule(static)
        for (size_t c0 = 0; c0 < half; c0 += CH) {
            u64 e[1] = {(u64)c0};
            fe_pow(F, tw[c0], wn, e, 1);
            size_t hi = c0 + CH < half ? c0 + CH : half;
            for (size_t j = c0 + 1; j < hi; j++) fe_mul(F, tw[j], tw[j - 1], wn);
        }
    }
    for (unsigned st = 1; st <= log_n; st++) {
        const size_t h = (size_t)1 << (st - 1), stride = half / h;
        #pragma omp parallel for schedule(static)
        for (size_t idx = 0; idx < half; idx++) {
            const size_t j = idx & (h - 1), lo = ((idx >> (st - 1)) << st) + j;
            u64 t[MAXL], u[MAXL];
            fe_mul(F, t, a[lo + h], tw[j * stride]); memcpy(u, a[lo], 32);
            fe_add(F, a[lo], u, t); fe_sub(F, a[lo + h], u, t);
        }
    }
    free(tw);
    if (inverse) {
        u64 nn[MAXL] = {(u64)n}, ni[MAXL];
        fe_to_mont(F, ni, nn); fe_inv(F, ni, ni);
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) fe_mul(F, a[i], a[i], ni);
    }
    memcpy(out, a, n * 32); free(a);
    return 0;
}

/* ---- MSM ---------------------------------------------------------------------------------------------------- */
/* pTSizes (min.js:1@212827): Pippenger window width by log2(n) */
static const int PT_SIZES[32] = {1,1,1,1,2,3,4,5,6,7,7,8,9,10,11,12,13,13,14,15,16,16,17,17,17,17,17,17,17,17,17,17};
static int ilog2(size_t n) { int l = 0; while (((size_t)1 << (l + 1)) <= n) l++; return l; }   /* ffjavascript log2(): floor */
static unsigned get_bits(const uint8_t *s, int nbytes, int start, int len) {
    unsigned v = 0;
    for (int b = 0; b < len; b++) { int bit = start + b; if (bit < 8 * nbytes && ((s[bit / 8] >> (bit % 8)) & 1)) v |= 1u << b; }
    return v;
}
/* One Pippenger window = g1m_multiexpAffine_chunk (min.js:1@75966): bucket[d] += P_i for digit d != 0, then
 * sum_d d·bucket[d] (the reference's _reduceTable @74634 does this by recursive halving; a running sum yields the
 * same group element). */
static void msm_window(const ext *E, u64 *R, const uint8_t *bases, const uint8_t *scalars, size_t n, int sb, int start, int c) {
    const int L = E->L; size_t nb = (size_t)1 << c;
    u64 *bk = calloc(nb * 3 * L, 8);
    u64 A[2 * MAXE];
    for (size_t i = 0; i < n; i++) {
        unsigned d = get_bits(scalars + i * sb, sb, start, c);
        if (!d) continue;
        memcpy(A, bases + i * 16 * L, 16 * L);
        if (aff_is_zero(E, A)) continue;
        pt_add_affine(E, bk + d * 3 * L, bk + d * 3 * L, A);
    }
    u64 run[3 * MAXE], acc[3 * MAXE]; pt_zero(E, run); pt_zero(E, acc);
    for (size_t d = nb - 1; d >= 1; d--) { pt_add(E, run, run, bk + d * 3 * L); pt_add(E, acc, acc, run); }
    memcpy(R, acc, 8 * 3 * L); free(bk);
}
/* G.multiExpAffine(bases, scalars) (driver min.js:1@214651 -> _multiExpChunk @213360): c = pTSizes[log2 n],
 * nWin = floor((8·sb-1)/c)+1 unsigned windows, recombined high -> low with c doublings each. The reference also splits
 * n into index chunks over workers and adds the partial results; addition is associative, so one chunk suffices here.
 * Output: Jacobian M, 3·n8q·group bytes; zero -> all-zero bytes. Scalars are plain integers of sb bytes, NOT reduced. */
int orc_msm(int curve, int group, const uint8_t *bases, const uint8_t *scalars, size_t n, int sb, uint8_t *out) {
    curve_t *C = get_curve(curve); ext E = make_ext(C, group); const int L = E.L;
    u64 res[3 * MAXE]; pt_zero(&E, res);
    if (n) {
        int c = PT_SIZES[ilog2(n)], nwin = (8 * sb - 1) / c + 1;
        u64 (*parts)[3 * MAXE] = malloc((size_t)nwin * sizeof(*parts));
        /* the window sums are independent (the reference hands them to its workers as separate tasks, @213360) */
        #pragma omp parallel for schedule(dynamic, 1)
        for (int w = 0; w < nwin; w++) {
            int len = 8 * sb - w * c; if (len > c) len = c;
            msm_window(&E, parts[w], bases, scalars, n, sb, w * c, len);
        }
        for (int w = nwin - 1; w >= 0; w--) {
            if (!pt_is_zero(&E, res)) for (int k = 0; k < c; k++) pt_double(&E, res, res);
            pt_add(&E, res, res, parts[w]);
        }
        free(parts);
    }
    if (pt_is_zero(&E, res)) memset(out, 0, 8 * 3 * L); else memcpy(out, res, 8 * 3 * L);
    return 0;
}
/* independent cross-check: plain double-and-add per term */
int orc_msm_naive(int curve, int group, const uint8_t *bases, const uint8_t *scalars, size_t n, int sb, uint8_t *out) {
    curve_t *C = get_curve(curve); ext E = make_ext(C, group); const int L = E.L;
    u64 res[3 * MAXE], P[3 * MAXE], T[3 * MAXE], A[2 * MAXE]; pt_zero(&E, res);
    for (size_t i = 0; i < n; i++) {
        memcpy(A, bases + i * 16 * L, 16 * L); pt_from_affine(&E, P, A);
        pt_mul_bytes(&E, T, P, scalars + i * sb, sb); pt_add(&E, res, res, T);
    }
    if (pt_is_zero(&E, res)) memset(out, 0, 8 * 3 * L); else memcpy(out, res, 8 * 3 * L);
    return 0;
}
/* G.toAffine on one Jacobian point -> affine M (zero -> all-zero bytes) */
int orc_to_affine(int curve, int group, const uint8_t *jac, uint8_t *aff) {
    curve_t *C = get_curve(curve); ext E = make_ext(C, group); u64 P[3 * MAXE], A[2 * MAXE];
    memcpy(P, jac, 8 * 3 * E.L); pt_to_affine(&E, A, P); memcpy(aff, A, 8 * 2 * E.L); return 0;
}
int orc_point_eq(int curve, int group, const uint8_t *jac_a, const uint8_t *jac_b) {
    uint8_t a[16 * MAXE], b[16 * MAXE]; int L = make_ext(get_curve(curve), group).L;
    orc_to_affine(curve, group, jac_a, a); orc_to_affine(curve, group, jac_b, b);
    return memcmp(a, b, 16 * L) == 0;
}
/* R = k·G (k plain LE bytes), Jacobian out — for the closed-form MSM check  sum s_i·7·11^i mod r · G */
int orc_generator_mul(int curve, int group, const uint8_t *k, int nb, uint8_t *out_jac) {
    curve_t *C = get_curve(curve); ext E = make_ext(C, group); u64 P[3 * MAXE], R[3 * MAXE];
    pt_from_affine(&E, P, group == 1 ? C->g1 : C->g2); pt_mul_bytes(&E, R, P, k, nb); memcpy(out_jac, R, 8 * 3 * E.L); return 0;
}
/* Synthetic base table of SURVEY.md §8d / Appendix C.1: P_i = (7·11^i mod r)·G, affine M — what
 * G.batchApplyKey(G repeated n, Fr.e(7), Fr.e(11)) returns. Built incrementally (P_{i+1} = 11·P_i) and
 * normalised with one batched inversion. */
int orc_geom_bases(int curve, int group, size_t n, uint8_t *out) {
    curve_t *C = get_curve(curve); ext E = make_ext(C, group); const int L = E.L;
    if (!n) return 0;
    u64 *J = malloc(n * 3 * L * 8), *pre = malloc(n * L * 8);
    u64 G[3 * MAXE]; uint8_t seven = 7, eleven = 11;
    pt_from_affine(&E, G, group == 1 ? C->g1 : C->g2);
    pt_mul_bytes(&E, J, G, &seven, 1);
    for (size_t i = 1; i < n; i++) pt_mul_bytes(&E, J + i * 3 * L, J + (i - 1) * 3 * L, &eleven, 1);
    u64 acc[MAXE], inv[MAXE], zi[MAXE], zi2[MAXE], zi3[MAXE], A[2 * MAXE];
    e_one(&E, acc);
    for (size_t i = 0; i < n; i++) { memcpy(pre + i * L, acc, 8 * L); e_mul(&E, acc, acc, J + i * 3 * L + 2 * L); }
    e_inv(&E, inv, acc);
    for (size_t i = n; i-- > 0;) {
        e_mul(&E, zi, inv, pre + i * L); e_mul(&E, inv, inv, J + i * 3 * L + 2 * L);
        e_sqr(&E, zi2, zi); e_mul(&E, zi3, zi2, zi);
        e_mul(&E, A, J + i * 3 * L, zi2); e_mul(&E, A + L, J + i * 3 * L + L, zi3);
        memcpy(out + i * 16 * L, A, 16 * L);
    }
    free(J); free(pre);
    return 0;
}

/* ---- Groth16 prover stages (src/groth16_prove.js) --------------------------------------------------------- */
static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* buildABC1 (src/groth16_prove.js:147-187): coeffs = zkey section 4 = u32 nCoef, then nCoef × {u32 matrix, u32
 * constraint, u32 signal, Fr value (stored ×R^2, src/zkey_utils.js:174-179)}; witness = wtns section 2 (normal form).
 * A_T[c] += coef·w[s] (matrix 0), B_T likewise (matrix 1) — Montgomery mul of (v·R^2) with plain w gives (v·w)·R —
 * then C_T = A_T ∘ B_T. Outputs 3 × n × 32 B (M). */
int orc_groth16_build_abc(int curve, const uint8_t *coeffs, size_t coeffs_len, const uint8_t *witness, size_t n_vars,
                          size_t domain, uint8_t *outA, uint8_t *outB, uint8_t *outC) {
    curve_t *C = get_curve(curve); const fld *F = &C->Fr;
    const size_t sCoef = 12 + 32, nCoef = (coeffs_len - 4) / sCoef;
    memset(outA, 0, domain * 32); memset(outB, 0, domain * 32);
    uint8_t *ob[2] = {outA, outB};
    for (size_t i = 0; i < nCoef; i++) {
        const uint8_t *rec = coeffs + 4 + i * sCoef;
        uint32_t m = rd32(rec), c = rd32(rec + 4), s = rd32(rec + 8);
        if (m > 1 || c >= domain || s >= n_vars) return -1;
        u64 cf[MAXL], w[MAXL], acc[MAXL], t[MAXL];
        memcpy(cf, rec + 12, 32); memcpy(w, witness + 32 * (size_t)s, 32); memcpy(acc, ob[m] + 32 * (size_t)c, 32);
        fe_mul(F, t, cf, w); fe_add(F, acc, acc, t); memcpy(ob[m] + 32 * (size_t)c, acc, 32);
    }
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < domain; i++) {
        u64 a[MAXL], b[MAXL], c[MAXL];
        memcpy(a, outA + 32 * i, 32); memcpy(b, outB + 32 * i, 32); fe_mul(F, c, a, b); memcpy(outC + 32 * i, c, 32);
    }
    return 0;
}
/* joinABC (src/groth16_prove.js:320-374): qap_joinABC (min.js:1@123664) P[i] = A[i]·B[i] − C[i] in Montgomery form,
 * then frm_batchFromMontgomery → normal form (these are the H-MSM scalars). */
int orc_groth16_join_abc(int curve, const uint8_t *A, const uint8_t *B, const uint8_t *Cc, size_t n, uint8_t *out) {
    curve_t *C = get_curve(curve); const fld *F = &C->Fr;
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 a[MAXL], b[MAXL], c[MAXL], t[MAXL];
        memcpy(a, A + 32 * i, 32); memcpy(b, B + 32 * i, 32); memcpy(c, Cc + 32 * i, 32);
        fe_mul(F, t, a, b); fe_sub(F, t, t, c); fe_from_mont(F, t, t); memcpy(out + 32 * i, t, 32);
    }
    return 0;
}
/* timesFr(P, k): k is an M Fr element; the reference converts it to a plain integer first. */
static void pt_times_fr(const curve_t *C, const ext *E, u64 *R, const u64 *P, const u64 *k_mont) {
    u64 k[MAXL]; fe_from_mont(&C->Fr, k, k_mont); pt_mul_bytes(E, R, P, (const uint8_t *)k, 32);
}
/* groth16Prove (src/groth16_prove.js:28-144), from the already-read sections:
 *   coeffs = zkey §4, bases A,B1,C,H = §5,§6,§8,§9 (G1 affine M), B2 = §7 (G2 affine M), witness = wtns §2,
 *   vk points = zkey §2 header points (affine M), r_mont / s_mont = the two Fr.random() draws (M form).
 * Outputs pi_a, pi_c (G1 affine M, 2·n8q) and pi_b (G2 affine M, 4·n8q); proof JSON = fromMontgomery of these. */
int orc_groth16_prove(int curve, size_t n_vars, size_t n_public, size_t domain, const uint8_t *coeffs, size_t coeffs_len,
                      const uint8_t *witness, const uint8_t *basesA, const uint8_t *basesB1, const uint8_t *basesB2,
                      const uint8_t *basesC, const uint8_t *basesH, const uint8_t *vk_alpha1, const uint8_t *vk_beta1,
                      const uint8_t *vk_beta2, const uint8_t *vk_delta1, const uint8_t *vk_delta2, const uint8_t *r_mont,
                      const uint8_t *s_mont, uint8_t *pi_a, uint8_t *pi_b, uint8_t *pi_c) {
    curve_t *C = get_curve(curve); const fld *Fr = &C->Fr;
    ext E1 = make_ext(C, 1), E2 = make_ext(C, 2);
    const int n8q = 8 * C->Fq.n;
    unsigned power = (unsigned)ilog2(domain);
    if (((size_t)1 << power) != domain) return -1;
    uint8_t *A = malloc(domain * 32), *B = malloc(domain * 32), *Cc = malloc(domain * 32), *T = malloc(domain * 32), *P = malloc(domain * 32);
    int rc = orc_groth16_build_abc(curve, coeffs, coeffs_len, witness, n_vars, domain, A, B, Cc);
    if (rc) { free(A); free(B); free(Cc); free(T); free(P); return rc; }
    /* inc = power == Fr.s ? Fr.shift : Fr.w[power+1]  (:64); Fr.shift = nqr^2 */
    u64 inc[MAXL], one[MAXL];
    memcpy(one, Fr->one, 32);
    if ((int)power == C->s) { u64 k[MAXL] = {25}; fe_to_mont(Fr, inc, k); /* nqr = 5 on both curves (golden-checked) */ }
    else memcpy(inc, C->w[power + 1], 32);
    uint8_t *bufs[3] = {A, B, Cc};
    for (int k = 0; k < 3; k++) {                                   /* :66-76 */
        orc_fr_ntt(curve, bufs[k], T, power, 1);
        orc_fr_batch_apply_key(curve, T, T, domain, (const uint8_t *)one, (const uint8_t *)inc);
        orc_fr_ntt(curve, T, bufs[k], power, 0);
    }
    orc_groth16_join_abc(curve, A, B, Cc, domain, P);                /* :79 */
    uint8_t ja[48 * 3], jb1[48 * 3], jb2[96 * 3], jc[48 * 3], jh[48 * 3];
    orc_msm(curve, 1, basesA, witness, n_vars, 32, ja);              /* :85 */
    orc_msm(curve, 1, basesB1, witness, n_vars, 32, jb1);            /* :89 */
    orc_msm(curve, 2, basesB2, witness, n_vars, 32, jb2);            /* :93 */
    orc_msm(curve, 1, basesC, witness + (n_public + 1) * 32, n_vars - n_public - 1, 32, jc);   /* :97 */
    orc_msm(curve, 1, basesH, P, domain, 32, jh);                    /* :101 */
    free(A); free(B); free(Cc); free(T); free(P);
    u64 pa[3 * MAXE], pb[3 * MAXE], pb1[3 * MAXE], pc[3 * MAXE], ph[3 * MAXE], t[3 * MAXE], q[3 * MAXE];
    u64 r[MAXL], s[MAXL], rs[MAXL], a1[2 * MAXE], b1[2 * MAXE], b2[2 * MAXE], d1[2 * MAXE], d2[2 * MAXE];
    memcpy(pa, ja, 3 * n8q); memcpy(pb1, jb1, 3 * n8q); memcpy(pb, jb2, 6 * n8q); memcpy(pc, jc, 3 * n8q); memcpy(ph, jh, 3 * n8q);
    memcpy(r, r_mont, 32); memcpy(s, s_mont, 32);
    memcpy(a1, vk_alpha1, 2 * n8q); memcpy(b1, vk_beta1, 2 * n8q); memcpy(b2, vk_beta2, 4 * n8q); memcpy(d1, vk_delta1, 2 * n8q); memcpy(d2, vk_delta2, 4 * n8q);
    pt_add_affine(&E1, pa, pa, a1);                                  /* :106 pi_a = A + alpha */
    pt_from_affine(&E1, q, d1); pt_times_fr(C, &E1, t, q, r); pt_add(&E1, pa, pa, t);       /* :107 + r·delta1 */
    pt_add_affine(&E2, pb, pb, b2);                                  /* :109 */
    pt_from_affine(&E2, q, d2); pt_times_fr(C, &E2, t, q, s); pt_add(&E2, pb, pb, t);       /* :110 */
    pt_add_affine(&E1, pb1, pb1, b1);                                /* :112 */
    pt_from_affine(&E1, q, d1); pt_times_fr(C, &E1, t, q, s); pt_add(&E1, pb1, pb1, t);     /* :113 */
    pt_add(&E1, pc, pc, ph);                                         /* :115 */
    pt_times_fr(C, &E1, t, pa, s); pt_add(&E1, pc, pc, t);           /* :118 */
    pt_times_fr(C, &E1, t, pb1, r); pt_add(&E1, pc, pc, t);          /* :119 */
    fe_mul(Fr, rs, r, s); fe_neg(Fr, rs, rs);
    pt_from_affine(&E1, q, d1); pt_times_fr(C, &E1, t, q, rs); pt_add(&E1, pc, pc, t);      /* :120 */
    u64 A2[2 * MAXE];
    pt_to_affine(&E1, A2, pa); memcpy(pi_a, A2, 2 * n8q);            /* :130-132 */
    pt_to_affine(&E2, A2, pb); memcpy(pi_b, A2, 4 * n8q);
    pt_to_affine(&E1, A2, pc); memcpy(pi_c, A2, 2 * n8q);
    return 0;
}
/* Fq element from Montgomery to normal form (for rendering proof coordinates as decimal strings) */
int orc_fq_from_mont(int curve, const uint8_t *in, uint8_t *out, size_t n) {
    curve_t *C = get_curve(curve); const int nb = 8 * C->Fq.n; u64 x[MAXL], y[MAXL];
    for (size_t i = 0; i < n; i++) { memcpy(x, in + nb * i, nb); fe_from_mont(&C->Fq, y, x); memcpy(out + nb * i, y, nb); }
    return 0;
}

static void pt_times_fr(const curve_t *C, const ext *E, u64 *R, const u64 *P, const u64 *k_mont);
/* ---- group-element FFT and batchApplyKey -----------------------------------------------------------------
 * G.fft / G.ifft (engine_fft for G1 / G2, min.js:1@215859 with g1m_/g2m_fftMix/_fftJoin/_fftFinal): X_k = sum_j w^(jk) P_j with
 * w = Fr.w[log n], natural order, the inverse scaled by 1/n; "multiply by a twiddle" = G.timesFr. Restated as bit-reversal +
 * radix-2 decimation in time. Affine (Montgomery, all-zero = infinity) in and out. */
int orc_group_fft(int curve, int group, const uint8_t *in, uint8_t *out, unsigned log_n, int inverse) {
    curve_t *C = get_curve(curve); const fld *Fr = &C->Fr; ext E = make_ext(C, group);
    if ((int)log_n > C->s) return -1;
    const size_t n = (size_t)1 << log_n, PB = (size_t)16 * E.L, JW = (size_t)3 * E.L;
    u64 *a = malloc(n * JW * 8);
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        size_t r = 0; for (unsigned b = 0; b < log_n; b++) r |= ((i >> b) & 1) << (log_n - 1 - b);
        u64 A[2 * MAXE]; memcpy(A, in + i * PB, PB); pt_from_affine(&E, a + r * JW, A);
    }
    for (unsigned st = 1; st <= log_n; st++) {
        const size_t h = (size_t)1 << (st - 1);
        const u64 *wst = inverse ? C->wi[st] : C->w[st];
        #pragma omp parallel for schedule(dynamic, 16)
        for (size_t idx = 0; idx < n / 2; idx++) {
            const size_t j = idx & (h - 1), lo = ((idx >> (st - 1)) << st) + j;
            u64 tw[MAXL], e[1] = {(u64)j}, T[3 * MAXE], U[3 * MAXE], N[3 * MAXE];
            fe_pow(Fr, tw, wst, e, 1);
            pt_times_fr(C, &E, T, a + (lo + h) * JW, tw);
            memcpy(U, a + lo * JW, JW * 8);
            pt_add(&E, a + lo * JW, U, T);
            pt_neg(&E, N, T);
            pt_add(&E, a + (lo + h) * JW, U, N);
        }
    }
    u64 ni[MAXL] = {0};
    if (inverse) { u64 nn[MAXL] = {(u64)n}; fe_to_mont(Fr, ni, nn); fe_inv(Fr, ni, ni); }
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; i++) {
        u64 P[3 * MAXE], A[2 * MAXE];
        if (inverse) pt_times_fr(C, &E, P, a + i * JW, ni); else memcpy(P, a + i * JW, JW * 8);
        pt_to_affine(&E, A, P); memcpy(out + i * PB, A, PB);
    }
    free(a);
    return 0;
}
/* G.batchApplyKey(buff, first, inc) (engine_applykey, min.js:1@211529; g1m_/g2m_batchApplyKey): out_i = (first * inc^i) * P_i, affine */
int orc_group_apply_key(int curve, int group, const uint8_t *in, uint8_t *out, size_t n, const uint8_t *first, const uint8_t *inc) {
    curve_t *C = get_curve(curve); const fld *Fr = &C->Fr; ext E = make_ext(C, group);
    const size_t PB = (size_t)16 * E.L;
    u64 f0[MAXL], k[MAXL];
    memcpy(f0, first, 32); memcpy(k, inc, 32);
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; i++) {
        u64 t[MAXL], e[1] = {(u64)i}, A[2 * MAXE], P[3 * MAXE], Q[3 * MAXE];
        fe_pow(Fr, t, k, e, 1); fe_mul(Fr, t, t, f0);
        memcpy(A, in + i * PB, PB); pt_from_affine(&E, P, A);
        pt_times_fr(C, &E, Q, P, t);
        pt_to_affine(&E, A, Q); memcpy(out + i * PB, A, PB);
    }
    return 0;
}

/* ---- helpers of the full-size closed-form checks (tests/test_gpu_parity.py) ------------------------------------------
 * sum_i s_i * f * g^i mod r over the entries with (i % skip_mod) != skip_rem (skip_mod = 0: all entries); s_i are plain
 * little-endian integers of sb <= 32 bytes (not reduced), result in normal form (32 bytes LE). These are the discrete logs
 * of MSM results over the geometric base table P_i = f*g^i*G. */
int orc_fr_geom_dot(int curve, const uint8_t *scalars, size_t n, int sb, u64 f, u64 g, unsigned skip_mod, unsigned skip_rem, uint8_t *out) {
    curve_t *C = get_curve(curve); const fld *F = &C->Fr;
    if (sb < 1 || sb > 32) return -1;
    u64 fm[MAXL] = {f}, gm[MAXL] = {g};
    fe_to_mont(F, fm, fm); fe_to_mont(F, gm, gm);
    const size_t CH = 8192, nch = (n + CH - 1) / CH;
    u64 (*part)[4] = calloc(nch ? nch : 1, 32);
    #pragma omp parallel for schedule(static)
    for (size_t ci = 0; ci < nch; ci++) {
        const size_t c0 = ci * CH, hi = c0 + CH < n ? c0 + CH : n;
        u64 t[MAXL], x[MAXL], acc[MAXL] = {0}, e[1] = {(u64)c0};
        fe_pow(F, t, gm, e, 1); fe_mul(F, t, t, fm);
        for (size_t i = c0; i < hi; i++) {
            if (!(skip_mod && (i % skip_mod) == skip_rem)) {
                memset(x, 0, sizeof x); memcpy(x, scalars + i * (size_t)sb, (size_t)sb);
                fe_to_mont(F, x, x);                   /* any 256-bit integer -> its residue in M form */
                fe_mul(F, x, x, t); fe_add(F, acc, acc, x);
            }
            fe_mul(F, t, t, gm);
        }
        memcpy(part[ci], acc, 32);
    }
    u64 acc[MAXL] = {0};
    for (size_t ci = 0; ci < nch; ci++) fe_add(F, acc, acc, part[ci]);
    free(part);
    fe_from_mont(F, acc, acc);
    memcpy(out, acc, 32);
    return 0;
}
/* element-wise Fr vector helpers for the valid-key synthesiser of the tests (tests/synth_valid_groth16.py): op 0 add, 1 sub, 2 mul;
 * Montgomery in, Montgomery out */
int orc_fr_vec_op(int curve, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {
    curve_t *C = get_curve(curve); const fld *F = &C->Fr;
    if (op < 0 || op > 2) return -1;
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 x[MAXL], y[MAXL], z[MAXL];
        memcpy(x, a + 32 * i, 32); memcpy(y, b + 32 * i, 32);
        if (op == 0) fe_add(F, z, x, y); else if (op == 1) fe_sub(F, z, x, y); else fe_mul(F, z, x, y);
        memcpy(out + 32 * i, z, 32);
    }
    return 0;
}
/* sum_i a_i * w_i mod r with a_i in Montgomery form and w_i plain 32-byte integers (not reduced): result in normal form */
int orc_fr_dot(int curve, const uint8_t *a_mont, const uint8_t *w_plain, size_t n, uint8_t *out) {
    curve_t *C = get_curve(curve); const fld *F = &C->Fr;
    const size_t CH = 8192, nch = (n + CH - 1) / CH;
    u64 (*part)[4] = calloc(nch ? nch : 1, 32);
    #pragma omp parallel for schedule(static)
    for (size_t ci = 0; ci < nch; ci++) {
        const size_t c0 = ci * CH, hi = c0 + CH < n ? c0 + CH : n;
        u64 acc[MAXL] = {0}, x[MAXL], y[MAXL], z[MAXL];
        for (size_t i = c0; i < hi; i++) {
            memcpy(x, a_mont + 32 * i, 32); memcpy(y, w_plain + 32 * i, 32);
            fe_mul(F, z, x, y);                        /* (a R)(w) R^-1 = a w, any w < 2^256 */
            fe_add(F, acc, acc, z);
        }
        memcpy(part[ci], acc, 32);
    }
    u64 acc[MAXL] = {0};
    for (size_t ci = 0; ci < nch; ci++) fe_add(F, acc, acc, part[ci]);
    free(part);
    memcpy(out, acc, 32);
    return 0;
}
/* ---- point-format conversions of the ceremony files -------------------------------------------------------
 * G.batchLEMtoU / batchUtoLEM / batchLEMtoC / batchCtoLEM (wasmcurves g1m_/g2m_batchLEMtoU ..., reached through ffjavascript's
 * engine_batchconvert, min.js:1@128060; callers src/powersoftau_import.js:159,221, src/powersoftau_contribute.js:145,176,
 * src/mpc_applykey.js:64-70, src/zkey_export_bellman.js:36-83, src/zkey_new.js:103-115). The byte formats are pinned by
 * tests/golden/<curve>_conv_* (outputs of the reference, oracle/gen_golden.js convertVectors):
 *   LEM  affine, little-endian, Montgomery; infinity = all zero
 *   U    affine, BIG-endian, normal form, x || y; an Fq2 coordinate is written c1 || c0; infinity = all zero
 *   C    x only, big-endian normal form (Fq2: c1 || c0); first byte |= 0x80 when y is "negative" (y > (p-1)/2; for Fq2 the
 *        test is made on c1, on c0 when c1 = 0); infinity = 0x40 followed by zeros
 * CtoLEM recovers y = sqrt(x^3 + b) (b = 3 / 4 on G1, 3/(9+u) / 4(1+u) on the twists) and picks the root whose sign matches the flag.
 * kind: 0 LEMtoU, 1 UtoLEM, 2 LEMtoC, 3 CtoLEM. Returns -2 when a compressed x has no point on the curve. */
static void be_store(uint8_t *out, const u64 *v, int n) { for (int i = 0; i < 8 * n; i++) out[i] = (uint8_t)(v[(8 * n - 1 - i) / 8] >> (8 * ((8 * n - 1 - i) % 8))); }
static void be_load(u64 *v, const uint8_t *in, int n) { memset(v, 0, 8 * n); for (int i = 0; i < 8 * n; i++) v[(8 * n - 1 - i) / 8] |= (u64)in[i] << (8 * ((8 * n - 1 - i) % 8)); }
static int fe_is_negative(const fld *F, const u64 *a_mont) {          /* normal form > (p-1)/2 */
    u64 a[MAXL], half[MAXL];
    fe_from_mont(F, a, a_mont);
    for (int i = 0; i < F->n; i++) half[i] = (F->p[i] >> 1) | (i + 1 < F->n ? F->p[i + 1] << 63 : 0);
    return bn_cmp(a, half, F->n) > 0;
}
static int e_is_negative(const ext *E, const u64 *a) {
    const int n = E->F->n;
    if (E->deg == 1) return fe_is_negative(E->F, a);
    return bn_is_zero(a + n, n) ? fe_is_negative(E->F, a) : fe_is_negative(E->F, a + n);
}
static int fe_sqrt(const fld *F, u64 *r, const u64 *a) {              /* p = 3 mod 4 on both curves: a^((p+1)/4), checked */
    u64 e[MAXL], one[MAXL] = {1}, t[MAXL];
    bn_add(e, F->p, one, F->n);
    for (int i = 0; i < F->n; i++) e[i] = (e[i] >> 2) | (i + 1 < F->n ? e[i + 1] << 62 : 0);
    fe_pow(F, r, a, e, F->n);
    fe_sqr(F, t, r);
    return memcmp(t, a, 8 * F->n) == 0;
}
static int e_sqrt(const ext *E, u64 *r, const u64 *a) {
    const fld *F = E->F; const int n = F->n;
    if (E->deg == 1) return fe_sqrt(F, r, a);
    /* norm method: (x0 + x1 u)^2 = a0 + a1 u  <=>  x0^2 = (a0 +- sqrt(a0^2 + a1^2)) / 2, x1 = a1 / (2 x0) */
    u64 t0[MAXL], t1[MAXL], s[MAXL], d[MAXL], x0[MAXL], x1[MAXL], inv2[MAXL], two[MAXL], chk[MAXE];
    if (bn_is_zero(a + n, n)) {
        if (fe_sqrt(F, r, a)) { memset(r + n, 0, 8 * n); return 1; }
        fe_neg(F, t0, a);
        if (!fe_sqrt(F, r + n, t0)) return 0;
        memset(r, 0, 8 * n); return 1;
    }
    fe_sqr(F, t0, a); fe_sqr(F, t1, a + n); fe_add(F, t0, t0, t1);
    if (!fe_sqrt(F, s, t0)) return 0;
    fe_add(F, two, F->one, F->one); fe_inv(F, inv2, two);
    fe_add(F, d, a, s); fe_mul(F, d, d, inv2);
    if (!fe_sqrt(F, x0, d)) { fe_sub(F, d, a, s); fe_mul(F, d, d, inv2); if (!fe_sqrt(F, x0, d)) return 0; }
    fe_add(F, t0, x0, x0); fe_inv(F, t0, t0); fe_mul(F, x1, a + n, t0);
    memcpy(r, x0, 8 * n); memcpy(r + n, x1, 8 * n);
    e_sqr(E, chk, r);
    return e_eq(E, chk, a);
}
static void curve_b(const curve_t *C, const ext *E, int curve, u64 *b) {   /* Montgomery form */
    const fld *F = &C->Fq; const int n = F->n;
    u64 k[MAXL] = {curve == 0 ? 3u : 4u}, km[MAXL];
    fe_to_mont(F, km, k);
    memset(b, 0, 8 * E->L);
    if (E->deg == 1) { memcpy(b, km, 8 * n); return; }
    if (curve == 0) {                                   /* 3 / (9 + u) */
        u64 xi[MAXE], nine[MAXL] = {9}, inv[MAXE], three[MAXE];
        fe_to_mont(F, xi, nine); memcpy(xi + n, F->one, 8 * n);
        e_inv(E, inv, xi);
        memset(three, 0, sizeof three); memcpy(three, km, 8 * n);
        e_mul(E, b, three, inv);
    } else { memcpy(b, km, 8 * n); memcpy(b + n, km, 8 * n); }          /* 4 (1 + u) */
}
int orc_group_convert(int curve, int group, int kind, const uint8_t *in, size_t n, uint8_t *out) {
    curve_t *C = get_curve(curve); const ext E = make_ext(C, group); const fld *F = &C->Fq;
    const int fn = F->n, nb = 8 * fn, sG = 2 * group * nb, sC = group * nb;
    u64 b[MAXE];
    curve_b(C, &E, curve, b);
    int bad = 0;
    #pragma omp parallel for schedule(static) reduction(|:bad)
    for (size_t i = 0; i < n; i++) {
        u64 P[2 * MAXE], t[MAXL];
        if (kind == 0 || kind == 2) {
            const uint8_t *src = in + i * sG; uint8_t *dst = out + i * (kind == 0 ? sG : sC);
            memcpy(P, src, sG);
            const int inf = bn_is_zero(P, 2 * E.L);
            for (int c = 0; c < (kind == 0 ? 2 : 1); c++)
                for (int k = 0; k < group; k++) { fe_from_mont(F, t, P + (c * group + k) * fn); be_store(dst + (c * group + (group - 1 - k)) * nb, t, fn); }
            if (kind == 2) { if (inf) dst[0] |= 0x40; else if (e_is_negative(&E, P + E.L)) dst[0] |= 0x80; }
        } else if (kind == 1) {
            const uint8_t *src = in + i * sG;
            for (int c = 0; c < 2; c++)
                for (int k = 0; k < group; k++) { be_load(t, src + (c * group + (group - 1 - k)) * nb, fn); fe_to_mont(F, P + (c * group + k) * fn, t); }
            memcpy(out + i * sG, P, sG);
        } else {
            uint8_t x[2 * 8 * MAXL];
            memcpy(x, in + i * sC, sC);
            const int flags = x[0] & 0xc0; x[0] &= 0x3f;
            if (flags & 0x40) { memset(out + i * sG, 0, sG); continue; }
            for (int k = 0; k < group; k++) { be_load(t, x + (group - 1 - k) * nb, fn); fe_to_mont(F, P + k * fn, t); }
            u64 rhs[MAXE], y[MAXE];
            e_sqr(&E, rhs, P); e_mul(&E, rhs, rhs, P); e_add(&E, rhs, rhs, b);
            if (!e_sqrt(&E, y, rhs)) { bad |= 1; memset(out + i * sG, 0, sG); continue; }
            if (e_is_negative(&E, y) != ((flags & 0x80) != 0)) e_neg(&E, y, y);
            memcpy(P + E.L, y, 8 * E.L);
            memcpy(out + i * sG, P, sG);
        }
    }
    return bad ? -2 : 0;
}

#ifdef _OPENMP
#include <omp.h>
int orc_threads(void) { return omp_get_max_threads(); }
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
int orc_threads(void) { return 1; }
void orc_set_threads(int n) { (void)n; }
#endif
