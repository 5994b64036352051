// snarkjs_amd/csrc/host_field.hpp — host-side (CPU) Montgomery arithmetic and Jacobian point ops.
//
// Product-side host code: used for O(1)/O(log n) work that brackets the HIP kernels — twiddle/power tables,
// the final fold of per-window MSM results (the reference also recombines windows on the host: ffjavascript
// _multiExpChunk, build/snarkjs.min.js:1@213360), Groth16 blinding (src/groth16_prove.js:103-132).
// NOT a fallback for the bulk kernels: there is no CPU path for MSM/NTT in the product.
#pragma once
#include <stdint.h>
#include <string.h>

namespace zkmi {
namespace host {

typedef unsigned __int128 u128;

// L = number of 64-bit limbs (4 or 6)
template <int L> struct HFp {
    uint64_t v[L];
    bool is_zero() const { uint64_t o = 0; for (int i = 0; i < L; i++) o |= v[i]; return o == 0; }
    bool operator==(const HFp& b) const { return memcmp(v, b.v, sizeof v) == 0; }
};

template <int L> struct HField {
    uint64_t p[L], one[L], r2[L], np;
    typedef HFp<L> E;

    template <class Cfg> static HField from_cfg() {
        static_assert(Cfg::N == 2 * L, "limb mismatch");
        HField F;
        for (int i = 0; i < L; i++) {
            F.p[i] = (uint64_t)Cfg::p(2 * i) | ((uint64_t)Cfg::p(2 * i + 1) << 32);
            F.one[i] = (uint64_t)Cfg::one(2 * i) | ((uint64_t)Cfg::one(2 * i + 1) << 32);
            F.r2[i] = (uint64_t)Cfg::r2(2 * i) | ((uint64_t)Cfg::r2(2 * i + 1) << 32);
        }
        uint64_t inv = 1;
        for (int i = 0; i < 6; i++) inv *= 2 - F.p[0] * inv;
        F.np = (uint64_t)0 - inv;
        return F;
    }
    static int cmp(const uint64_t* a, const uint64_t* b) {
        for (int i = L - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
        return 0;
    }
    E zero() const { E r; memset(r.v, 0, sizeof r.v); return r; }
    E One() const { E r; memcpy(r.v, one, sizeof r.v); return r; }
    E R2() const { E r; memcpy(r.v, r2, sizeof r.v); return r; }
    E add(const E& a, const E& b) const {
        E r; u128 c = 0;
        for (int i = 0; i < L; i++) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
        if (c || cmp(r.v, p) >= 0) { uint64_t bw = 0; for (int i = 0; i < L; i++) { u128 d = (u128)r.v[i] - p[i] - bw; r.v[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1; } }
        return r;
    }
    E sub(const E& a, const E& b) const {
        E r; uint64_t bw = 0;
        for (int i = 0; i < L; i++) { u128 d = (u128)a.v[i] - b.v[i] - bw; r.v[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1; }
        if (bw) { u128 c = 0; for (int i = 0; i < L; i++) { c += (u128)r.v[i] + p[i]; r.v[i] = (uint64_t)c; c >>= 64; } }
        return r;
    }
    E neg(const E& a) const { return a.is_zero() ? a : sub(zero(), a); }
    E dbl(const E& a) const { return add(a, a); }
    // CIOS Montgomery multiplication, 64-bit limbs
    E mul(const E& a, const E& b) const {
        uint64_t t[L + 2];
        memset(t, 0, sizeof t);
        for (int i = 0; i < L; i++) {
            u128 c = 0;
            for (int j = 0; j < L; j++) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[L]; t[L] = (uint64_t)c; t[L + 1] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * np;
            c = ((u128)m * p[0] + t[0]) >> 64;
            for (int j = 1; j < L; j++) { c += (u128)m * p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[L]; t[L - 1] = (uint64_t)c; t[L] = t[L + 1] + (uint64_t)(c >> 64);
        }
        E r; memcpy(r.v, t, sizeof r.v);
        if (t[L] || cmp(r.v, p) >= 0) { uint64_t bw = 0; for (int i = 0; i < L; i++) { u128 d = (u128)r.v[i] - p[i] - bw; r.v[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1; } }
        return r;
    }
    E sqr(const E& a) const { return mul(a, a); }
    E to_mont(const E& a) const { return mul(a, R2()); }
    E from_mont(const E& a) const { E o = zero(); o.v[0] = 1; return mul(a, o); }
    E from_u64(uint64_t x) const { E o = zero(); o.v[0] = x; return to_mont(o); }
    E pow(const E& a, const uint64_t* e, int ne) const {
        E r = One(), b = a;
        for (int i = 0; i < 64 * ne; i++) { if ((e[i / 64] >> (i % 64)) & 1) r = mul(r, b); b = sqr(b); }
        return r;
    }
    E pow_u64(const E& a, uint64_t e) const { return pow(a, &e, 1); }
    E inv(const E& a) const {
        uint64_t e[L]; uint64_t bw = 2;
        for (int i = 0; i < L; i++) { e[i] = p[i] - bw; bw = p[i] < bw ? 1 : 0; }
        return pow(a, e, L);
    }
};

// Quadratic extension Fq2 = Fq[u]/(u^2+1)
template <int L> struct HFp2 { HFp<L> c0, c1; bool is_zero() const { return c0.is_zero() && c1.is_zero(); } bool operator==(const HFp2& b) const { return c0 == b.c0 && c1 == b.c1; } };
template <int L> struct HField2 {
    HField<L> F;
    typedef HFp2<L> E;
    E zero() const { return E{F.zero(), F.zero()}; }
    E One() const { return E{F.One(), F.zero()}; }
    E add(const E& a, const E& b) const { return E{F.add(a.c0, b.c0), F.add(a.c1, b.c1)}; }
    E sub(const E& a, const E& b) const { return E{F.sub(a.c0, b.c0), F.sub(a.c1, b.c1)}; }
    E neg(const E& a) const { return E{F.neg(a.c0), F.neg(a.c1)}; }
    E dbl(const E& a) const { return add(a, a); }
    E mul(const E& a, const E& b) const {
        auto t0 = F.mul(a.c0, b.c0), t1 = F.mul(a.c1, b.c1);
        auto t2 = F.mul(F.add(a.c0, a.c1), F.add(b.c0, b.c1));
        return E{F.sub(t0, t1), F.sub(F.sub(t2, t0), t1)};
    }
    E sqr(const E& a) const { return mul(a, a); }
    E inv(const E& a) const {
        auto d = F.inv(F.add(F.sqr(a.c0), F.sqr(a.c1)));
        return E{F.mul(a.c0, d), F.neg(F.mul(a.c1, d))};
    }
};
// uniform interface over Fq
template <int L> struct HField1 : HField<L> {
    HField1() {}
    HField1(const HField<L>& f) : HField<L>(f) {}
};

// Jacobian point over a field FT with element type FT::E; curve y^2 = x^3 + b (a = 0). Zero: Z = 0.
template <class FT> struct HPoint { typename FT::E X, Y, Z; };
template <class FT> struct HCurve {
    FT F;
    typedef typename FT::E E;
    typedef HPoint<FT> P;
    P zero() const { return P{F.zero(), F.zero(), F.zero()}; }
    bool is_zero(const P& p) const { return p.Z.is_zero(); }
    P from_affine(const E& x, const E& y) const { if (x.is_zero() && y.is_zero()) return zero(); return P{x, y, F.One()}; }
    P neg(const P& p) const { return P{p.X, F.neg(p.Y), p.Z}; }
    P dbl(const P& p) const {
        if (is_zero(p)) return p;
        E A = F.sqr(p.X), B = F.sqr(p.Y), C = F.sqr(B);
        E D = F.dbl(F.sub(F.sub(F.sqr(F.add(p.X, B)), A), C));
        E Ee = F.add(F.dbl(A), A), Ff = F.sqr(Ee);
        P r;
        r.X = F.sub(Ff, F.dbl(D));
        r.Y = F.sub(F.mul(Ee, F.sub(D, r.X)), F.dbl(F.dbl(F.dbl(C))));
        r.Z = F.dbl(F.mul(p.Y, p.Z));
        return r;
    }
    P add(const P& p, const P& q) const {
        if (is_zero(p)) return q;
        if (is_zero(q)) return p;
        E Z1Z1 = F.sqr(p.Z), Z2Z2 = F.sqr(q.Z);
        E U1 = F.mul(p.X, Z2Z2), U2 = F.mul(q.X, Z1Z1);
        E S1 = F.mul(F.mul(p.Y, q.Z), Z2Z2), S2 = F.mul(F.mul(q.Y, p.Z), Z1Z1);
        if (U1 == U2) return (S1 == S2) ? dbl(p) : zero();
        E H = F.sub(U2, U1), I = F.sqr(F.dbl(H)), J = F.mul(H, I);
        E r = F.dbl(F.sub(S2, S1)), V = F.mul(U1, I);
        P o;
        o.X = F.sub(F.sub(F.sqr(r), J), F.dbl(V));
        o.Y = F.sub(F.mul(r, F.sub(V, o.X)), F.dbl(F.mul(S1, J)));
        o.Z = F.mul(F.sub(F.sub(F.sqr(F.add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
        return o;
    }
    // k: plain little-endian integer, nbits wide
    P mul_bits(const P& p, const uint8_t* k, int nbits) const {
        P acc = zero();
        for (int i = nbits - 1; i >= 0; i--) { acc = dbl(acc); if ((k[i / 8] >> (i % 8)) & 1) acc = add(acc, p); }
        return acc;
    }
    void to_affine(const P& p, E& x, E& y) const {
        if (is_zero(p)) { x = F.zero(); y = F.zero(); return; }
        E zi = F.inv(p.Z), zi2 = F.sqr(zi);
        x = F.mul(p.X, zi2); y = F.mul(p.Y, F.mul(zi2, zi));
    }
};

}  // namespace host
}  // namespace zkmi
