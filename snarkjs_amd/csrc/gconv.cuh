// snarkjs_amd/csrc/gconv.cuh — point-format conversions of the ceremony files for gfx950.
//
// Replaces ffjavascript's engine_batchconvert over wasmcurves' g1m_/g2m_batchLEMtoU, _batchUtoLEM, _batchLEMtoC, _batchCtoLEM
// (build/snarkjs.min.js:1@128060; callers src/powersoftau_import.js:159,221, src/powersoftau_contribute.js:145,176,
// src/powersoftau_export_challenge.js:72, src/powersoftau_verify.js:358, src/mpc_applykey.js:64-70, src/zkey_export_bellman.js:36-83,
// src/zkey_new.js:103-115,373). Byte formats (pinned by tests/golden/<curve>_conv_*, outputs of the reference):
//   LEM  affine, little-endian, Montgomery, x || y; infinity = all zero
//   U    affine, big-endian, normal form, x || y, an Fq2 coordinate written c1 || c0; infinity = all zero
//   C    x alone, big-endian normal form; first byte |= 0x80 when y > (p-1)/2 (Fq2: decided on c1, on c0 when c1 = 0);
//        infinity = 0x40 then zeros
// LEM <-> U is byte work with one Montgomery product per field element: one lane per FIELD ELEMENT, so that a wavefront reads and
// writes contiguous 2-3 KiB (the element order inside an Fq2 coordinate is the only permutation). LEM -> C needs x and y of one point
// in one lane; C -> LEM is a square root per point (p = 3 mod 4 on both curves: one exponentiation in Fq, two in Fq2), ALU-bound.
#pragma once
#include "curve.cuh"

namespace zkmi {

ZK_DEV uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
// big-endian byte string of C::N words <-> little-endian limbs
template <class C> ZK_DEV Fp<C> fp_load_be(const uint32_t* p) {
    Fp<C> r, t = fp_load<C>(p);
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = bswap32(t.l[C::N - 1 - i]);
    return r;
}
template <class C> ZK_DEV void fp_store_be(uint32_t* p, const Fp<C>& a) {
    Fp<C> t;
#pragma unroll
    for (int i = 0; i < C::N; i++) t.l[i] = bswap32(a.l[C::N - 1 - i]);
    fp_store<C>(p, t);
}
// exponent (p + ADD) >> SHIFT, word i (ADD touches word 0 only: checked at compile time for the two moduli by the callers' static_assert)
template <class C, int ADD, int SHIFT> ZK_DEV uint32_t pexp_word(int i) {
    const uint32_t lo = (i == 0 ? (uint32_t)(C::p(0) + ADD) : C::p(i));
    const uint32_t hi = (i + 1 < C::N) ? C::p(i + 1) : 0u;
    return (lo >> SHIFT) | (hi << (32 - SHIFT));
}
template <class C, int ADD> constexpr bool pexp_ok() { return ADD >= 0 ? (C::p(0) + (uint32_t)ADD > C::p(0) || ADD == 0) : (C::p(0) >= (uint32_t)(-ADD)); }
// a^((p + ADD) >> SHIFT), MSB first; the exponent is the same in every lane
template <class F, class C, int ADD, int SHIFT> __device__ __noinline__ F f_pow_pexp(const F& a) {
    static_assert(pexp_ok<C, ADD>(), "exponent offset must stay inside word 0");
    F r;
    f_set_one(r);
    for (int w = C::N - 1; w >= 0; w--) {
        const uint32_t e = pexp_word<C, ADD, SHIFT>(w);
        for (int b = 31; b >= 0; b--) {
            r = f_sqr(r);
            if ((e >> b) & 1u) r = f_mul(r, a);
        }
    }
    return r;
}

// normal form > (p-1)/2
template <class C> ZK_DEV bool fp_is_negative(const Fp<C>& a_mont) {
    const Fp<C> a = fp_from_mont(a_mont);
    for (int i = C::N - 1; i >= 0; i--) {
        const uint32_t h = (C::p(i) >> 1) | ((i + 1 < C::N) ? (C::p(i + 1) << 31) : 0u);
        if (a.l[i] != h) return a.l[i] > h;
    }
    return false;
}
template <class C> ZK_DEV bool f_is_negative(const Fp<C>& a) { return fp_is_negative(a); }
template <class C> ZK_DEV bool f_is_negative(const Fp2<C>& a) { return fp_is_zero(a.c1) ? fp_is_negative(a.c0) : fp_is_negative(a.c1); }

// square roots; false when `a` is not a square
template <class C> ZK_DEV bool f_sqrt(Fp<C>& r, const Fp<C>& a) {
    r = f_pow_pexp<Fp<C>, C, 1, 2>(a);                               // a^((p+1)/4)
    return fp_eq(fp_sqr(r), a);
}
// Fq2 = Fq[u]/(u^2+1), p = 3 mod 4 (Adj, Rodriguez-Henriquez: "Square root computation over even extension fields", Alg. 9)
template <class C> ZK_DEV bool f_sqrt(Fp2<C>& r, const Fp2<C>& a) {
    const Fp2<C> a1 = f_pow_pexp<Fp2<C>, C, -3, 2>(a);              // a^((p-3)/4)
    const Fp2<C> x0 = f_mul(a1, a);
    const Fp2<C> alpha = f_mul(a1, x0);
    const Fp<C> minus_one = fp_neg(fp_one<C>());
    if (fp_is_zero(alpha.c1) && fp_eq(alpha.c0, minus_one)) r = Fp2<C>{fp_neg(x0.c1), x0.c0};         // u * x0
    else {
        Fp2<C> t = alpha;
        t.c0 = fp_add(t.c0, fp_one<C>());
        r = f_mul(f_pow_pexp<Fp2<C>, C, -1, 1>(t), x0);             // (1 + alpha)^((p-1)/2) * x0
    }
    return f_eq(f_sqr(r), a);
}

template <class F> struct BaseCfg;
template <class C> struct BaseCfg<Fp<C>> { typedef C type; static constexpr int DEG = 1; };
template <class C> struct BaseCfg<Fp2<C>> { typedef C type; static constexpr int DEG = 2; };
template <class C> ZK_DEV Fp<C>& f_comp(Fp<C>& a, int) { return a; }
template <class C> ZK_DEV Fp<C>& f_comp(Fp2<C>& a, int k) { return k ? a.c1 : a.c0; }

// LEM -> U (to_u) or U -> LEM: one lane per field element; `deg` = 1 (G1) or 2 (G2: c0/c1 swap places)
template <class C> __global__ void __launch_bounds__(256) k_gconv_elems(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n_elems, int deg, int to_u) {
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elems) return;
    const uint64_t o = (deg == 2) ? (e ^ 1ull) : e;
    if (to_u) fp_store_be<C>(out + o * C::N, fp_from_mont(fp_load<C>(in + e * C::N)));
    else fp_store<C>(out + o * C::N, fp_to_mont(fp_load_be<C>(in + e * C::N)));
}
// LEM -> C: one lane per point
template <class F> __global__ void __launch_bounds__(256) k_gconv_compress(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
    typedef typename BaseCfg<F>::type C;
    constexpr int D = BaseCfg<F>::DEG, FW = FieldWords<F>::value;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Affine<F> p;
    f_load(p.x, in + (size_t)i * 2 * FW);
    f_load(p.y, in + (size_t)i * 2 * FW + FW);
    const bool inf = pt_is_inf(p);
    const uint32_t flag = inf ? 0x40u : (f_is_negative(p.y) ? 0x80u : 0u);
    uint32_t* dst = out + (size_t)i * FW;
#pragma unroll
    for (int k = 0; k < D; k++) {
        const Fp<C> v = fp_from_mont(f_comp(p.x, D - 1 - k));
        Fp<C> t;
#pragma unroll
        for (int j = 0; j < C::N; j++) t.l[j] = bswap32(v.l[C::N - 1 - j]);
        if (k == 0) t.l[0] |= flag;                                   // first BYTE of the string = low byte of word 0
        fp_store<C>(dst + k * C::N, t);
    }
}
// C -> LEM: y = sqrt(x^3 + b), sign chosen by the flag; *bad |= 1 when some x has no point
template <class F> __global__ void __launch_bounds__(256) k_gconv_decompress(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ b_mont,
                                                                             uint32_t* __restrict__ bad) {
    typedef typename BaseCfg<F>::type C;
    constexpr int D = BaseCfg<F>::DEG, FW = FieldWords<F>::value;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = in + (size_t)i * FW;
    uint32_t* dst = out + (size_t)i * 2 * FW;
    const uint32_t flags = src[0] & 0xc0u;
    F x, y, b;
#pragma unroll
    for (int k = 0; k < D; k++) {
        Fp<C> t = fp_load<C>(src + k * C::N), v;
        if (k == 0) t.l[0] &= ~0xc0u;
#pragma unroll
        for (int j = 0; j < C::N; j++) v.l[j] = bswap32(t.l[C::N - 1 - j]);
        f_comp(x, D - 1 - k) = fp_to_mont(v);
    }
    f_load(b, b_mont);
    bool ok = true;
    if (flags & 0x40u) { f_set_zero(x); f_set_zero(y); }
    else {
        const F rhs = f_add(f_mul(f_sqr(x), x), b);
        ok = f_sqrt(y, rhs);
        if (!ok) { f_set_zero(x); f_set_zero(y); }
        else if (f_is_negative(y) != ((flags & 0x80u) != 0)) y = f_neg(y);
    }
    if (!ok) atomicOr(bad, 1u);
    f_store(dst, x);
    f_store(dst + FW, y);
}

}  // namespace zkmi
