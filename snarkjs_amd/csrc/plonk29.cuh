// snarkjs_amd/csrc/plonk29.cuh — PLONK's quotient numerator (computeT, src/plonk_prove.js:516-628 with MulZ.mul2 / mul4, src/mul_z.js:49-148)
// on unsaturated 9 x 29-bit limbs (field29.cuh). One evaluation point of the extended domain per call; plonk.hip wraps it in k_plonk_t29.
//
// Forms. The caller's bytes hold R-form values (x 2^256 mod r, canonical). A product of field29.cuh divides by R' = 2^261, so:
//   * an array element is loaded SHIFTED by 5 bits while it is cut into limbs (load29_shl: no arithmetic, the shifts of unpack29 with other
//     amounts): the limbs then hold 32 x 2^256 x = x 2^261 — the R'-form of x, lazily reduced (< 32 r). R'-form is closed under mul29;
//   * constants come twice, uploaded by the host: block k in R-form, block k29 in R'-form (both canonical);
//   * a product with ONE R-form operand (a selector polynomial's evaluation, alpha, L1) brings an R'-form value back to R-form — every
//     output below ends in such a product, so nothing is ever converted for its own sake.
// Sums of products share one Montgomery reduction (mul29_2 / _3 / _4): q_M ab + q_L a + q_R b + q_O c is one call, and the quadratic
// coefficients of MulZ.mul4 are taken directly — (A0 + A1 Z + A2 Z^2)(B0 + B1 Z + B2 Z^2) reduced by Z^2 -> Z1 Z, Z^3 -> Z2 Z, Z^4 -> Z3 Z is
//   rz = A0 (B1 + Z1 B2) + A1 (B0 + Z1 B1 + Z2 B2) + A2 (Z1 B0 + Z2 B1 + Z3 B2)
// = 29 product / reduction units of 81 multiply-adds for one mul4 (the reference spends 27 full multiplications, plonk.hip's Karatsuba form 15
// = 30 units plus 13 carry-chain additions). Z1 = Z2 = Z3 = 0 at the points with i % 4 == 0 (mul_z.js:21-47), so the reference's `if (p)` needs
// no branch: the same expression gives its value. Field arithmetic is exact — the stored words are the reference's, bit for bit.
//
// Bounds. Every function is written against an element type E through the traits E29<E>: E = Fp29<C> is the arithmetic; the host test
// (tools/plonk29_hosttest.hip) also runs the SAME body with E = an interval type that carries (value bound in units of r, limb bound, normalised?)
// and checks every precondition of field29.cuh — operand limbs of the products, column sums below 2^64, subtrahends under their offset, the
// ranges of the two final reductions — for BN254 Fr (R'/r = 169) and BLS12-381 Fr (R'/r = 70.7). Comments quote the BN254 / BLS12-381 value bounds.
#pragma once
#include "ntt29.cuh"

namespace zkmi {

// base^e = lo[e & (2^lb - 1)] * hi[e >> lb]   (tables of R-form words, built by plonk.hip: build_pow_tab)
struct PowTab {
    const uint32_t* lo;
    const uint32_t* hi;
    uint32_t lb;
};
// constants block: [0] beta [1] gamma [2] k1 [3] k2 [4] alpha [5] alpha^2 [6] w_n [7..17] b1..b11 [18..21] Z1 [22..25] Z2 [26..29] Z3 [30] one [31] -alpha
enum { PK_BETA = 0, PK_GAMMA, PK_K1, PK_K2, PK_ALPHA, PK_ALPHA2, PK_WN, PK_B1, PK_Z1 = PK_B1 + 11, PK_Z2 = PK_Z1 + 4, PK_Z3 = PK_Z2 + 4, PK_ONE = PK_Z3 + 4, PK_NALPHA, PK_COUNT };
struct PlonkTArgs {
    const uint32_t *a, *b, *c, *z, *qm, *ql, *qr, *qo, *qc, *s1, *s2, *s3;     // 4n evaluations each
    const uint32_t* lagrange;     // section 13 on the device: per public input 5n elements (n coefficients, 4n evaluations)
    const uint32_t* pub_a;        // buffers.A (Montgomery): A[j], j < nPublic
    const uint32_t* k;            // constants block, R-form
    const uint32_t* k29;          // the same constants in R'-form (x 2^261 mod r): read by the 29-bit kernels only
    uint32_t domain, n_public;
    uint32_t *t, *tz;
};

// N words holding v < 2^(32 N - S)  ->  limbs of v 2^S, S = r29_shift (5): limb k is bits [B k - S, B k - S + B) of v
template <class C> ZK_HD Fp29<C> load29_shl(const uint32_t* p) {
    constexpr int NL = Lim29<C>::NL, B = Lim29<C>::B, S = r29_shift<C>();
    static_assert(S > 0 && S < B, "shift inside the first limb");
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint32_t w[C::N];
#pragma unroll
    for (int i = 0; i < C::N / 4; i++) { const uint4 a = q[i]; w[4 * i] = a.x; w[4 * i + 1] = a.y; w[4 * i + 2] = a.z; w[4 * i + 3] = a.w; }
    Fp29<C> r;
    r.l[0] = (w[0] << S) & mask29<C>();
#pragma unroll
    for (int k = 1; k < NL; k++) {
        const int bit = B * k - S, wi = bit >> 5, sh = bit & 31;
        uint64_t v = w[wi];
        if (wi + 1 < C::N) v |= (uint64_t)w[wi + 1] << 32;
        r.l[k] = (k == NL - 1) ? (uint32_t)(v >> sh) : ((uint32_t)(v >> sh) & mask29<C>());
    }
#if defined(ZK29_SHADOW)
    b29::set(r, ldexp(1.0, S), b29::lowmax<C>(), -1.0);          // canonical words times 2^S
#endif
    return r;
}

// the sums of three and four products behind a call where the field is Compact<C> (field29.cuh: mul29 / mul29_2 already are)
template <class C> __host__ __device__ ZK_NOINLINE_DEV Fp29<C> mul29_3_call(Fp29<C> a0, Fp29<C> b0, Fp29<C> a1, Fp29<C> b1, Fp29<C> a2, Fp29<C> b2) { return mul29_3_inl(a0, b0, a1, b1, a2, b2); }
template <class C> __host__ __device__ ZK_NOINLINE_DEV Fp29<C> mul29_4_call(Fp29<C> a0, Fp29<C> b0, Fp29<C> a1, Fp29<C> b1, Fp29<C> a2, Fp29<C> b2, Fp29<C> a3, Fp29<C> b3) {
    return mul29_4(a0, b0, a1, b1, a2, b2, a3, b3);
}

// what the body needs from an element type
template <class E> struct E29;
template <class C> struct E29<Fp29<C>> {
    typedef Fp29<C> E;
    static ZK_HD E load(const uint32_t* p) { return load29_packed<C>(p); }                   // canonical words as they are: < r, normalised
    static ZK_HD E load_shl(const uint32_t* p) { return load29_shl<C>(p); }                  // 32 x the words: < 32 r, normalised
    static ZK_HD E zero() { return zero29<C>(); }
    static ZK_HD E one() { return one29<C>(); }                                              // 1 in R'-form
    static ZK_HD E mul(const E& a, const E& b) { return mul29(a, b); }
    static ZK_HD E mul2(const E& a0, const E& b0, const E& a1, const E& b1) { return mul29_2(a0, b0, a1, b1); }
    static ZK_HD E mul3(const E& a0, const E& b0, const E& a1, const E& b1, const E& a2, const E& b2) {
        if constexpr (IsCompact<C>::value) return mul29_3_call<C>(a0, b0, a1, b1, a2, b2); else return mul29_3_inl(a0, b0, a1, b1, a2, b2);
    }
    static ZK_HD E mul4(const E& a0, const E& b0, const E& a1, const E& b1, const E& a2, const E& b2, const E& a3, const E& b3) {
        if constexpr (IsCompact<C>::value) return mul29_4_call<C>(a0, b0, a1, b1, a2, b2, a3, b3); else return mul29_4(a0, b0, a1, b1, a2, b2, a3, b3);
    }
    static ZK_HD E add(const E& a, const E& b) { return add29(a, b); }
    static ZK_HD E sub2(const E& t, const E& b) { return sub29<C, 2>(t, b); }                // t + 2 r - b
    static ZK_HD void norm(E& a) { norm29(a); }
    static ZK_HD E reduce_lt32(E v) { norm29(v); reduce29_small(v); return v; }              // any lazy value below 32 r -> canonical
    static ZK_HD void put(uint32_t* dst, const E& v) {
        uint32_t w[C::N];
        pack29<C>(w, v);
        uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
        for (int i = 0; i < C::N / 4; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    }
    static ZK_HD void store_lt3(uint32_t* dst, E v) { norm29(v); canon29(v); put(dst, v); }  // lazy value below 3 r -> canonical words
    static ZK_HD void store_lt32(uint32_t* dst, E v) { put(dst, reduce_lt32(v)); }
};

// MulZ.mul4 (mul_z.js:103-148) in R'-form: (a + ap Z)(b + bp Z)(c + cp Z)(d + dp Z); every operand normalised. r: the constant coefficient,
// rz: the blinding part. Values (BN254 / BLS12-381): a, b, c < 34.1 / 34.5 r, d < 32 r, ap .. dp < 2.1 / 2.3 r.
template <class E> ZK_HD void mul4z29(const E& a, const E& b, const E& c, const E& d, const E& ap, const E& bp, const E& cp, const E& dp, const E& Z1, const E& Z2, const E& Z3, E& r, E& rz) {
    using X = E29<E>;
    const E A0 = X::mul(a, b), A1 = X::mul2(a, bp, ap, b), A2 = X::mul(ap, bp);              // < 7.9 / 17.8,  1.9 / 3.3,  1.03 / 1.08 r
    const E B0 = X::mul(c, d), B1 = X::mul2(c, dp, cp, d), B2 = X::mul(cp, dp);
    r = X::mul(A0, B0);                                                                       // < 1.4 / 5.2 r
    E c0 = X::add(B1, X::mul(Z1, B2)), c1 = X::add(B0, X::mul2(Z1, B1, Z2, B2));
    X::norm(c0); X::norm(c1);
    const E c2 = X::mul3(Z1, B0, Z2, B1, Z3, B2);
    rz = X::mul3(A0, c0, A1, c1, A2, c2);                                                     // < 1.3 / 2.9 r
}

// PART 0: t = e1 + e4, tz likewise (stores);  PART 1: t += alpha e2;  PART 2: t -= alpha e3   (plonk_prove.js:560-612; the split in three
// launches is plonk.hip's: the live set of all four terms together does not fit the register file at two waves per SIMD)
template <class E, int PART> ZK_HD void plonk_t29_point(const PlonkTArgs& g, const PowTab& w4, uint32_t i) {
    using X = E29<E>;
    const uint32_t n4 = 4 * g.domain, dom = g.domain;
    auto ld = [&](const uint32_t* p, size_t idx) { return X::load(p + idx * 8); };
    auto lds = [&](const uint32_t* p, size_t idx) { return X::load_shl(p + idx * 8); };
    auto kR = [&](int j) { return X::load(g.k + (size_t)j * 8); };          // constant j, R-form
    auto kS = [&](int j) { return X::load(g.k29 + (size_t)j * 8); };        // constant j, R'-form
    auto bl = [&](int j) { return kS(PK_B1 + j - 1); };                     // challenges.b[j]
    const E w = X::mul(lds(w4.lo, i & ((1u << w4.lb) - 1)), lds(w4.hi, i >> w4.lb));         // Fr.w[power+2]^i, R'-form, < 7.1 / 15.5 r
    E ap = X::add(bl(2), X::mul(bl(1), w)), bp = X::add(bl(4), X::mul(bl(3), w)), cp = X::add(bl(6), X::mul(bl(5), w));      // < 2.1 / 2.3 r
    X::norm(ap); X::norm(bp); X::norm(cp);
    const int zi = (int)(i & 3);
    uint32_t* pt = g.t + (size_t)i * 8;
    uint32_t* ptz = g.tz + (size_t)i * 8;
    if constexpr (PART == 0) {
        // e1 := a b qM + a qL + b qR + c qO + PI + qC ;  e4 := alpha^2 (z - 1) L1
        const E a = lds(g.a, i), b = lds(g.b, i), c = lds(g.c, i);
        const E qm = ld(g.qm, i), ql = ld(g.ql, i), qr = ld(g.qr, i), qo = ld(g.qo, i);      // R-form: the products below come out in R-form
        E pi = X::zero();
        for (uint32_t j = 0; j < g.n_public; j++) pi = X::reduce_lt32(X::sub2(pi, X::mul(lds(g.lagrange, (size_t)j * 5 * dom + dom + i), ld(g.pub_a, j))));
        const E e1 = X::mul4(X::mul(a, b), qm, a, ql, b, qr, c, qo);                          // < 1.7 / 2.6 r
        E rz = X::add(X::mul2(a, bp, ap, b), X::mul(kS(PK_Z1 + zi), X::mul(ap, bp)));         // MulZ.mul2: a bp + ap b + Z1 ap bp
        X::norm(rz);
        const E e1z = X::mul4(rz, qm, ap, ql, bp, qr, cp, qo);
        const E l1 = ld(g.lagrange, (size_t)dom + i), alpha2 = kS(PK_ALPHA2);
        const E e4 = X::mul(X::mul(X::sub2(lds(g.z, i), X::one()), l1), alpha2);              // (z - 1) L1 comes out in R-form; alpha^2 in R'-form keeps it there
        const E zp = X::add(X::mul(X::add(X::mul(bl(7), w), bl(8)), w), bl(9));
        const E e4z = X::mul(X::mul(zp, l1), alpha2);
        X::store_lt32(pt, X::add(X::add(e1, pi), X::add(ld(g.qc, i), e4)));                   // < 4.7 / 5.7 r
        X::store_lt3(ptz, X::add(e1z, e4z));                                                   // < 2.1 / 2.2 r
    } else {
        const E beta = kS(PK_BETA), gamma = kS(PK_GAMMA);
        E A, Bv, Cv, D, dp;
        if constexpr (PART == 1) {
            // e2 := alpha (a + beta X + gamma)(b + beta k1 X + gamma)(c + beta k2 X + gamma) z
            const E bw = X::mul(beta, w);
            A = X::add(X::add(lds(g.a, i), bw), gamma);
            Bv = X::add(X::add(lds(g.b, i), X::mul(bw, kS(PK_K1))), gamma);
            Cv = X::add(X::add(lds(g.c, i), X::mul(bw, kS(PK_K2))), gamma);
            D = lds(g.z, i);
            dp = X::add(X::mul(X::add(X::mul(bl(7), w), bl(8)), w), bl(9));
        } else {
            // e3 := alpha (a + beta s1 + gamma)(b + beta s2 + gamma)(c + beta s3 + gamma) z(X w)
            A = X::add(X::add(lds(g.a, i), X::mul(beta, lds(g.s1, i))), gamma);
            Bv = X::add(X::add(lds(g.b, i), X::mul(beta, lds(g.s2, i))), gamma);
            Cv = X::add(X::add(lds(g.c, i), X::mul(beta, lds(g.s3, i))), gamma);
            D = lds(g.z, (n4 + 4 + i) % n4);
            const E wW = X::mul(w, kS(PK_WN));
            dp = X::add(X::mul(X::add(X::mul(bl(7), wW), bl(8)), wW), bl(9));
        }
        X::norm(A); X::norm(Bv); X::norm(Cv); X::norm(dp);
        E r, rz;
        mul4z29<E>(A, Bv, Cv, D, ap, bp, cp, dp, kS(PK_Z1 + zi), kS(PK_Z2 + zi), kS(PK_Z3 + zi), r, rz);
        const E al = kR(PART == 1 ? PK_ALPHA : PK_NALPHA);                                     // R-form: t (R-form) + alpha e
        X::store_lt3(pt, X::add(ld(pt, 0), X::mul(r, al)));                                    // < 2.1 / 2.1 r
        X::store_lt3(ptz, X::add(ld(ptz, 0), X::mul(rz, al)));
    }
}

}  // namespace zkmi
