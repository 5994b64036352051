// snarkjs_amd/csrc/curve.cuh — short-Weierstrass (a = 0) group arithmetic for the MSM kernels, gfx950.
//
// Replaces wasmcurves' g1m_/g2m_ Jacobian add/addMixed/double (reference bundle build/snarkjs.min.js:1@86766) inside
// the Pippenger kernels. The accumulators use extended-Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2): a mixed addition costs 8M+2S (vs 7M+4S Jacobian) and a doubling never needs Z. Points cross the
// C-ABI in the reference's own formats: affine (x,y) Montgomery with all-zero = infinity in, Jacobian out.
#pragma once
#include "field.cuh"

namespace zkmi {

// ---- quadratic extension Fq2 = Fq[u]/(u^2+1) (both BN254 and BLS12-381) ----------------------------------------
template <class C> struct Fp2 {
    Fp<C> c0, c1;
    using Cfg = C;
};

// uniform free-function interface over Fp<C> and Fp2<C>
template <class C> ZK_DEV Fp<C> f_add(const Fp<C>& a, const Fp<C>& b) { return fp_add(a, b); }
template <class C> ZK_DEV Fp<C> f_sub(const Fp<C>& a, const Fp<C>& b) { return fp_sub(a, b); }
template <class C> ZK_DEV Fp<C> f_mul(const Fp<C>& a, const Fp<C>& b) { return fp_mul(a, b); }
template <class C> ZK_DEV Fp<C> f_sqr(const Fp<C>& a) { return fp_sqr(a); }
template <class C> ZK_DEV Fp<C> f_dbl(const Fp<C>& a) { return fp_dbl(a); }
template <class C> ZK_DEV Fp<C> f_neg(const Fp<C>& a) { return fp_neg(a); }
template <class C> ZK_DEV bool f_is_zero(const Fp<C>& a) { return fp_is_zero(a); }
template <class C> ZK_DEV bool f_eq(const Fp<C>& a, const Fp<C>& b) { return fp_eq(a, b); }
template <class C> ZK_DEV void f_set_zero(Fp<C>& a) { a = fp_zero<C>(); }
template <class C> ZK_DEV void f_set_one(Fp<C>& a) { a = fp_one<C>(); }
template <class C> ZK_DEV Fp<C> f_inv(const Fp<C>& a) { return fp_inv(a); }

template <class C> ZK_DEV Fp2<C> f_add(const Fp2<C>& a, const Fp2<C>& b) { return Fp2<C>{fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
template <class C> ZK_DEV Fp2<C> f_sub(const Fp2<C>& a, const Fp2<C>& b) { return Fp2<C>{fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
template <class C> ZK_DEV Fp2<C> f_dbl(const Fp2<C>& a) { return Fp2<C>{fp_dbl(a.c0), fp_dbl(a.c1)}; }
template <class C> ZK_DEV Fp2<C> f_neg(const Fp2<C>& a) { return Fp2<C>{fp_neg(a.c0), fp_neg(a.c1)}; }
template <class C> ZK_DEV bool f_is_zero(const Fp2<C>& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
template <class C> ZK_DEV bool f_eq(const Fp2<C>& a, const Fp2<C>& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
template <class C> ZK_DEV void f_set_zero(Fp2<C>& a) { a.c0 = fp_zero<C>(); a.c1 = fp_zero<C>(); }
template <class C> ZK_DEV void f_set_one(Fp2<C>& a) { a.c0 = fp_one<C>(); a.c1 = fp_zero<C>(); }
// Base-field product used inside Fq2 arithmetic. An Fq2 mixed addition holds 28 of them: fully inlined that is > 64 KiB
// of straight-line code (the instruction cache two CUs share) and > 200 VGPRs; as an out-of-line call (operands and
// result in registers) the hot loop stays cache-resident.  ZKMI_FP2_NOINLINE=0 restores full inlining.
#ifndef ZKMI_FP2_NOINLINE
#define ZKMI_FP2_NOINLINE 1
#endif
#if ZKMI_FP2_NOINLINE == 1
template <class C> __device__ __noinline__ Fp<C> fp2_base_mul(Fp<C> a, Fp<C> b) { return fp_mul(a, b); }
#else
template <class C> ZK_DEV Fp<C> fp2_base_mul(const Fp<C>& a, const Fp<C>& b) { return fp_mul(a, b); }
#endif
#if ZKMI_FP2_NOINLINE == 2
// the call boundary sits at the Fq2 operation: 32 argument registers in, 16 out, the Karatsuba partial products never live
// across a call (the caller's live set across any call of a mixed addition is <= 4 Fq2 values: fits the callee-saved VGPRs)
#define ZK_FP2_OP __device__ __noinline__
#define ZK_FP2_ARG(T) T
#else
#define ZK_FP2_OP ZK_DEV
#define ZK_FP2_ARG(T) const T&
#endif
// Karatsuba: 3 base-field multiplications
template <class C> ZK_FP2_OP Fp2<C> f_mul(ZK_FP2_ARG(Fp2<C>) a, ZK_FP2_ARG(Fp2<C>) b) {
    Fp<C> t0 = fp2_base_mul(a.c0, b.c0), t1 = fp2_base_mul(a.c1, b.c1);
    Fp<C> t2 = fp2_base_mul(fp_add_noreduce(a.c0, a.c1), fp_add_noreduce(b.c0, b.c1));
    return Fp2<C>{fp_sub(t0, t1), fp_sub(fp_sub(t2, t0), t1)};
}
// (a0+a1)(a0-a1) + 2 a0 a1 u : 2 multiplications
template <class C> ZK_FP2_OP Fp2<C> f_sqr(ZK_FP2_ARG(Fp2<C>) a) {
    Fp<C> t = fp2_base_mul(a.c0, a.c1);
    return Fp2<C>{fp2_base_mul(fp_add_noreduce(a.c0, a.c1), fp_sub(a.c0, a.c1)), fp_dbl(t)};
}
template <class C> ZK_DEV Fp2<C> f_inv(const Fp2<C>& a) {
    Fp<C> d = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
    return Fp2<C>{fp_mul(a.c0, d), fp_neg(fp_mul(a.c1, d))};
}

// element size in 32-bit words
template <class F> struct FieldWords;
template <class C> struct FieldWords<Fp<C>> { static constexpr int value = C::N; };
template <class C> struct FieldWords<Fp2<C>> { static constexpr int value = 2 * C::N; };

template <class C> ZK_DEV void f_load(Fp<C>& r, const uint32_t* p) { r = fp_load<C>(p); }
template <class C> ZK_DEV void f_store(uint32_t* p, const Fp<C>& a) { fp_store<C>(p, a); }
template <class C> ZK_DEV void f_load(Fp2<C>& r, const uint32_t* p) { r.c0 = fp_load<C>(p); r.c1 = fp_load<C>(p + C::N); }
template <class C> ZK_DEV void f_store(uint32_t* p, const Fp2<C>& a) { fp_store<C>(p, a.c0); fp_store<C>(p + C::N, a.c1); }

// ---- points ----------------------------------------------------------------------------------------------------
template <class F> struct Affine { F x, y; };                 // infinity: x = y = 0 (the reference's all-zero encoding)
template <class F> struct XYZZ { F X, Y, ZZ, ZZZ; };           // infinity: ZZ = 0

template <class F> ZK_DEV bool pt_is_inf(const Affine<F>& p) { return f_is_zero(p.x) && f_is_zero(p.y); }
template <class F> ZK_DEV bool pt_is_inf(const XYZZ<F>& p) { return f_is_zero(p.ZZ); }
template <class F> ZK_DEV void pt_set_inf(XYZZ<F>& p) { f_set_zero(p.X); f_set_zero(p.Y); f_set_zero(p.ZZ); f_set_zero(p.ZZZ); }

template <class F> ZK_DEV void pt_load(Affine<F>& p, const uint32_t* src) {
    constexpr int W = FieldWords<F>::value;
    f_load(p.x, src); f_load(p.y, src + W);
}
template <class F> ZK_DEV void pt_load(XYZZ<F>& p, const uint32_t* src) {
    constexpr int W = FieldWords<F>::value;
    f_load(p.X, src); f_load(p.Y, src + W); f_load(p.ZZ, src + 2 * W); f_load(p.ZZZ, src + 3 * W);
}
template <class F> ZK_DEV void pt_store(uint32_t* dst, const XYZZ<F>& p) {
    constexpr int W = FieldWords<F>::value;
    f_store(dst, p.X); f_store(dst + W, p.Y); f_store(dst + 2 * W, p.ZZ); f_store(dst + 3 * W, p.ZZZ);
}

// doubling of an affine point into XYZZ (EFD mdbl-2008-s-1, a = 0)
template <class F> ZK_DEV XYZZ<F> pt_dbl_affine(const Affine<F>& p) {
    XYZZ<F> r;
    F U = f_dbl(p.y), V = f_sqr(U), W = f_mul(U, V), S = f_mul(p.x, V);
    F xx = f_sqr(p.x), M = f_add(f_dbl(xx), xx);
    r.X = f_sub(f_sqr(M), f_dbl(S));
    r.Y = f_sub(f_mul(M, f_sub(S, r.X)), f_mul(W, p.y));
    r.ZZ = V; r.ZZZ = W;
    return r;
}
// XYZZ doubling (EFD dbl-2008-s-1, a = 0). noinline: used off the hot loop (bucket reduction); keeps code size and
// compile time bounded — the call costs nothing next to 9 field multiplications.
template <class F> __device__ __noinline__ XYZZ<F> pt_dbl(const XYZZ<F>& p) {
    if (pt_is_inf(p)) return p;
    XYZZ<F> r;
    F U = f_dbl(p.Y), V = f_sqr(U), W = f_mul(U, V), S = f_mul(p.X, V);
    F xx = f_sqr(p.X), M = f_add(f_dbl(xx), xx);
    r.X = f_sub(f_sqr(M), f_dbl(S));
    r.Y = f_sub(f_mul(M, f_sub(S, r.X)), f_mul(W, p.Y));
    r.ZZ = f_mul(V, p.ZZ); r.ZZZ = f_mul(W, p.ZZZ);
    return r;
}
// acc += q (q affine, not infinity unless flagged by the caller). EFD madd-2008-s: 8M + 2S.
template <class F> ZK_DEV void pt_madd(XYZZ<F>& acc, const Affine<F>& q) {
    if (pt_is_inf(q)) return;
    if (pt_is_inf(acc)) { acc.X = q.x; acc.Y = q.y; f_set_one(acc.ZZ); f_set_one(acc.ZZZ); return; }
    F U2 = f_mul(q.x, acc.ZZ), S2 = f_mul(q.y, acc.ZZZ);
    F P = f_sub(U2, acc.X), R = f_sub(S2, acc.Y);
    if (f_is_zero(P)) {
        if (f_is_zero(R)) acc = pt_dbl_affine(q); else pt_set_inf(acc);
        return;
    }
    F PP = f_sqr(P), PPP = f_mul(P, PP), Q = f_mul(acc.X, PP);
    F X3 = f_sub(f_sub(f_sqr(R), PPP), f_dbl(Q));
    acc.Y = f_sub(f_mul(R, f_sub(Q, X3)), f_mul(acc.Y, PPP));
    acc.X = X3;
    acc.ZZ = f_mul(acc.ZZ, PP); acc.ZZZ = f_mul(acc.ZZZ, PPP);
}
// ---- accumulator parked in LDS -----------------------------------------------------------------------------------------
// An Fq2 XYZZ accumulator is 64 (BN254) / 96 (BLS12-381) VGPRs; together with the operands of a mixed addition the live set
// exceeds 256 registers and the compiler spills to scratch (measured: 15 GB of scratch writes per 2^20 G2 MSM). Instead the
// four coordinates live in LDS (transposed: dword i of lane t at i*T + t, conflict-free) and are pulled into registers only
// while needed.  T = threads per block.
template <class C> ZK_DEV void f_from_words(Fp<C>& v, const uint32_t* w) {
#pragma unroll
    for (int i = 0; i < C::N; i++) v.l[i] = w[i];
}
template <class C> ZK_DEV void f_to_words(uint32_t* w, const Fp<C>& v) {
#pragma unroll
    for (int i = 0; i < C::N; i++) w[i] = v.l[i];
}
template <class C> ZK_DEV void f_from_words(Fp2<C>& v, const uint32_t* w) {
#pragma unroll
    for (int i = 0; i < C::N; i++) { v.c0.l[i] = w[i]; v.c1.l[i] = w[C::N + i]; }
}
template <class C> ZK_DEV void f_to_words(uint32_t* w, const Fp2<C>& v) {
#pragma unroll
    for (int i = 0; i < C::N; i++) { w[i] = v.c0.l[i]; w[C::N + i] = v.c1.l[i]; }
}
template <class F, int T> struct LdsAcc {
    static constexpr int FW = FieldWords<F>::value;
    uint32_t* base;                         // &lds[threadIdx.x]
    ZK_DEV void get(int coord, F& v) const {
        uint32_t w[FW];
#pragma unroll
        for (int i = 0; i < FW; i++) w[i] = base[(coord * FW + i) * T];
        f_from_words(v, w);
    }
    ZK_DEV void put(int coord, const F& v) const {
        uint32_t w[FW];
        f_to_words(w, v);
#pragma unroll
        for (int i = 0; i < FW; i++) base[(coord * FW + i) * T] = w[i];
    }
};
// acc += q (q affine and not infinity); `inf` = accumulator is the point at infinity (kept in a register)
template <class F, int T> ZK_DEV void pt_madd_lds(const LdsAcc<F, T>& A, bool& inf, const Affine<F>& q) {
    if (inf) { F one; f_set_one(one); A.put(0, q.x); A.put(1, q.y); A.put(2, one); A.put(3, one); inf = false; return; }
    F t, P, R;
    A.get(2, t); P = f_mul(q.x, t);            // U2
    A.get(0, t); P = f_sub(P, t);              // P = U2 - X
    A.get(3, t); R = f_mul(q.y, t);            // S2
    A.get(1, t); R = f_sub(R, t);              // R = S2 - Y
    if (f_is_zero(P)) {
        if (f_is_zero(R)) { XYZZ<F> d = pt_dbl_affine(q); A.put(0, d.X); A.put(1, d.Y); A.put(2, d.ZZ); A.put(3, d.ZZZ); }
        else inf = true;
        return;
    }
    F PP = f_sqr(P);
    A.get(2, t); A.put(2, f_mul(t, PP));       // ZZ' = ZZ*PP
    A.get(0, t);
    F Q = f_mul(t, PP);                        // Q = X*PP
    F PPP = f_mul(P, PP);
    A.get(3, t); A.put(3, f_mul(t, PPP));      // ZZZ' = ZZZ*PPP
    F X3 = f_sub(f_sub(f_sqr(R), PPP), f_dbl(Q));
    A.put(0, X3);
    A.get(1, t);
    A.put(1, f_sub(f_mul(R, f_sub(Q, X3)), f_mul(t, PPP)));
}
// Doubling of an LDS-parked accumulator in place (EFD dbl-2008-s-1, a = 0): the rare equal-points branch of pt_add_lds.
template <class F, int T> ZK_DEV void pt_dbl_lds(const LdsAcc<F, T>& A) {
    F t, U, V, Wv, S, M;
    A.get(1, t); U = f_dbl(t); V = f_sqr(U); Wv = f_mul(U, V);
    F YW = f_mul(Wv, t);                        // W*Y
    A.get(0, t); S = f_mul(t, V);
    t = f_sqr(t); M = f_add(f_dbl(t), t);
    F X3 = f_sub(f_sqr(M), f_dbl(S));
    A.put(0, X3);
    A.put(1, f_sub(f_mul(M, f_sub(S, X3)), YW));
    A.get(2, t); A.put(2, f_mul(V, t));
    A.get(3, t); A.put(3, f_mul(Wv, t));
}
// acc (LDS-parked) += p for a general XYZZ point p whose coordinates are fetched on demand: ld(coord, F&) with coord 0..3 =
// X, Y, ZZ, ZZZ (from global memory or from another lane's LDS accumulator). EFD add-2008-s, 12M + 2S, all special cases.
// Neither operand ever sits in registers as a whole: the live set stays within 256 VGPRs with no scratch — the by-reference
// pt_add below keeps both Fq2 points plus the result in a scratch frame (768 B per lane), which made every Fq2 addition of the
// bucket reduction ~300 us of latency.
template <class F, int T, class Ld> ZK_DEV void pt_add_lds(const LdsAcc<F, T>& A, bool& inf, Ld ld) {
    F t, u;
    ld(2, u);                                   // ZZ2
    if (f_is_zero(u)) return;                   // p = infinity
    if (inf) {
        A.put(2, u);
        ld(0, t); A.put(0, t); ld(1, t); A.put(1, t); ld(3, t); A.put(3, t);
        inf = false;
        return;
    }
    F P, R, U1, S1;
    A.get(0, t); U1 = f_mul(t, u);              // U1 = X1*ZZ2
    A.get(2, t); ld(0, u); P = f_sub(f_mul(u, t), U1);      // P = X2*ZZ1 - U1
    A.get(1, t); ld(3, u); S1 = f_mul(t, u);    // S1 = Y1*ZZZ2
    A.get(3, t); ld(1, u); R = f_sub(f_mul(u, t), S1);      // R = Y2*ZZZ1 - S1
    if (f_is_zero(P)) {
        if (f_is_zero(R)) pt_dbl_lds(A); else inf = true;
        return;
    }
    F PP = f_sqr(P);
    A.get(2, t); ld(2, u); A.put(2, f_mul(f_mul(t, u), PP));      // ZZ3 = ZZ1*ZZ2*PP
    F Q = f_mul(U1, PP);
    F PPP = f_mul(P, PP);
    A.get(3, t); ld(3, u); A.put(3, f_mul(f_mul(t, u), PPP));     // ZZZ3 = ZZZ1*ZZZ2*PPP
    F X3 = f_sub(f_sub(f_sqr(R), PPP), f_dbl(Q));
    A.put(0, X3);
    A.put(1, f_sub(f_mul(R, f_sub(Q, X3)), f_mul(S1, PPP)));
}
// full addition (EFD add-2008-s: 12M + 2S) with all special cases; inlined flavour for throughput-bound loops
template <class F> ZK_DEV XYZZ<F> pt_add_inl(const XYZZ<F>& a, const XYZZ<F>& b) {
    if (pt_is_inf(a)) return b;
    if (pt_is_inf(b)) return a;
    F U1 = f_mul(a.X, b.ZZ), U2 = f_mul(b.X, a.ZZ), S1 = f_mul(a.Y, b.ZZZ), S2 = f_mul(b.Y, a.ZZZ);
    F P = f_sub(U2, U1), R = f_sub(S2, S1);
    if (f_is_zero(P)) {
        if (f_is_zero(R)) return pt_dbl(a);
        XYZZ<F> z; pt_set_inf(z); return z;
    }
    XYZZ<F> r;
    F PP = f_sqr(P), PPP = f_mul(P, PP), Q = f_mul(U1, PP);
    r.X = f_sub(f_sub(f_sqr(R), PPP), f_dbl(Q));
    r.Y = f_sub(f_mul(R, f_sub(Q, r.X)), f_mul(S1, PPP));
    r.ZZ = f_mul(f_mul(a.ZZ, b.ZZ), PP);
    r.ZZZ = f_mul(f_mul(a.ZZZ, b.ZZZ), PPP);
    return r;
}
// out-of-line flavour for latency-bound code (LDS trees, scans): keeps code size and compile time bounded
template <class F> __device__ __noinline__ XYZZ<F> pt_add(const XYZZ<F>& a, const XYZZ<F>& b) {
    if (pt_is_inf(a)) return b;
    if (pt_is_inf(b)) return a;
    F U1 = f_mul(a.X, b.ZZ), U2 = f_mul(b.X, a.ZZ), S1 = f_mul(a.Y, b.ZZZ), S2 = f_mul(b.Y, a.ZZZ);
    F P = f_sub(U2, U1), R = f_sub(S2, S1);
    if (f_is_zero(P)) {
        if (f_is_zero(R)) return pt_dbl(a);
        XYZZ<F> z; pt_set_inf(z); return z;
    }
    XYZZ<F> r;
    F PP = f_sqr(P), PPP = f_mul(P, PP), Q = f_mul(U1, PP);
    r.X = f_sub(f_sub(f_sqr(R), PPP), f_dbl(Q));
    r.Y = f_sub(f_mul(R, f_sub(Q, r.X)), f_mul(S1, PPP));
    r.ZZ = f_mul(f_mul(a.ZZ, b.ZZ), PP);
    r.ZZZ = f_mul(f_mul(a.ZZZ, b.ZZZ), PPP);
    return r;
}

}  // namespace zkmi
