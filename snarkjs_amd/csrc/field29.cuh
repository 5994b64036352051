// snarkjs_amd/csrc/field29.cuh — unsaturated-limb Montgomery arithmetic for the MSM accumulation kernels (gfx950).
//
// A 254-bit field element as 9 limbs of 29 bits, Montgomery factor R' = 2^261. The point: in product scanning a column holds
// up to 18 products of < 2^58, which fits a 64-bit accumulator — so every MAC is ONE v_mad_u64_u32, where the saturated
// 8 x 32-bit form (field.cuh) needs MAC + carry counter. Measured (tools/fieldbench29.hip, profiles/r02_fieldbench29.txt):
// 175 Gmul/s against 130 Gmul/s for the same modulus, 150 against 116 at 2 waves per SIMD.
//
// Additions and subtractions are LAZY: no carry propagation and no reduction mod p unless asked for.
//   * value invariant: every element is < 2^258 (R'/p = 2^7.4 leaves the headroom; a product of two such values divided by R'
//     comes back below 1.6 p);
//   * limb invariant: a NORMALISED element has limbs 0..7 < 2^29 (the top limb carries the excess of the value); mul29 accepts
//     one operand with limbs < 2^31 when the other is normalised, or two operands with limbs < 2^30; subtrahends must be normalised;
//   * a - b is computed as a - b + K p with K p in a redundant form whose limbs dominate any normalised subtrahend whose value is
//     below (K p - 2^232): K is chosen per call site from the value bounds written next to it (curve29 in msm29.cuh).
// Memory formats: the reference's bytes (8 x 32-bit words, Montgomery factor 2^256, canonical) at every kernel boundary; window
// tables of resident bases are kept as canonical 8-word values in R'-form (x * 2^261 mod p), so a gathered point is unpacked with
// shifts only.
#pragma once
#include "field.cuh"

namespace zkmi {

constexpr uint32_t M29 = (1u << 29) - 1;

// per-modulus constants (generated with Python big integers; checked at start-up against host arithmetic: msm29 self-test)
template <class C> struct Lim29;
template <> struct Lim29<Bn254Fq> {
    static constexpr uint32_t NP = 0x04866389u;                    // -p^-1 mod 2^29
    static constexpr uint32_t PINV = 0x1b799c77u;                  //  p^-1 mod 2^29
    ZK_HD static constexpr uint32_t p(int i) { constexpr uint32_t v[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu}; return v[i]; }
    ZK_HD static constexpr uint32_t one(int i) { constexpr uint32_t v[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u}; return v[i]; }     // 2^261 mod p
    ZK_HD static constexpr uint32_t kin(int i) { constexpr uint32_t v[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u}; return v[i]; }     // 2^266 mod p: R-form -> R'-form
    ZK_HD static constexpr uint32_t kout(int i) { constexpr uint32_t v[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u}; return v[i]; }    // 2^256 mod p: R'-form -> R-form
};

template <class C> struct Fp29 {
    uint32_t l[9];
    using Cfg = C;
};

ZK_DEV void mad29(uint64_t& acc, uint32_t a, uint32_t b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
ZK_DEV void mad29c(uint64_t& acc, uint32_t a, uint32_t k) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc"); }

// a * b / 2^261 mod p, lazily reduced (< a b / R' + p), normalised limbs
template <class C> ZK_DEV Fp29<C> mul29(const Fp29<C>& a, const Fp29<C>& b) {
    using L = Lim29<C>;
    Fp29<C> r;
    uint32_t m[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mad29(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mad29c(acc, m[i], L::p(k - i));
        m[k] = ((uint32_t)acc * L::NP) & M29;
        mad29c(acc, m[k], L::p(0));
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) mad29(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 8; i < 9; i++) mad29c(acc, m[i], L::p(k - i));
        r.l[k - 9] = (k == 17) ? (uint32_t)acc : ((uint32_t)acc & M29);
        acc >>= 29;
    }
    return r;
}
// (a0 * b0 + a1 * b1) / 2^261 mod p with ONE reduction (all four operands normalised): the two components of an Fq2 product without
// Karatsuba's operand sums and without its subtractions (the caller passes b1 already negated where the formula has a minus)
template <class C> ZK_DEV Fp29<C> mul29_2(const Fp29<C>& a0, const Fp29<C>& b0, const Fp29<C>& a1, const Fp29<C>& b1) {
    using L = Lim29<C>;
    Fp29<C> r;
    uint32_t m[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { mad29(acc, a0.l[i], b0.l[k - i]); mad29(acc, a1.l[i], b1.l[k - i]); }
#pragma unroll
        for (int i = 0; i < k; i++) mad29c(acc, m[i], L::p(k - i));
        m[k] = ((uint32_t)acc * L::NP) & M29;
        mad29c(acc, m[k], L::p(0));
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) { mad29(acc, a0.l[i], b0.l[k - i]); mad29(acc, a1.l[i], b1.l[k - i]); }
#pragma unroll
        for (int i = k - 8; i < 9; i++) mad29c(acc, m[i], L::p(k - i));
        r.l[k - 9] = (k == 17) ? (uint32_t)acc : ((uint32_t)acc & M29);
        acc >>= 29;
    }
    return r;
}
// (a0 b0 + a1 b1 + a2 b2 + a3 b3) / 2^261 mod p with ONE reduction: every operand normalised (limbs < 2^29), so that a column holds at most
// 36 products of < 2^58 plus 9 reduction terms: < 2^63.5. Used where a formula ends in a difference of Fq2 products (Y3 = R T - Y1 PPP).
template <class C> ZK_DEV Fp29<C> mul29_4(const Fp29<C>& a0, const Fp29<C>& b0, const Fp29<C>& a1, const Fp29<C>& b1, const Fp29<C>& a2, const Fp29<C>& b2, const Fp29<C>& a3,
                                          const Fp29<C>& b3) {
    using L = Lim29<C>;
    Fp29<C> r;
    uint32_t m[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { mad29(acc, a0.l[i], b0.l[k - i]); mad29(acc, a1.l[i], b1.l[k - i]); mad29(acc, a2.l[i], b2.l[k - i]); mad29(acc, a3.l[i], b3.l[k - i]); }
#pragma unroll
        for (int i = 0; i < k; i++) mad29c(acc, m[i], L::p(k - i));
        m[k] = ((uint32_t)acc * L::NP) & M29;
        mad29c(acc, m[k], L::p(0));
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) { mad29(acc, a0.l[i], b0.l[k - i]); mad29(acc, a1.l[i], b1.l[k - i]); mad29(acc, a2.l[i], b2.l[k - i]); mad29(acc, a3.l[i], b3.l[k - i]); }
#pragma unroll
        for (int i = k - 8; i < 9; i++) mad29c(acc, m[i], L::p(k - i));
        r.l[k - 9] = (k == 17) ? (uint32_t)acc : ((uint32_t)acc & M29);
        acc >>= 29;
    }
    return r;
}
// a^2 / 2^261 mod p for a normalised a: 45 products instead of 81 — every cross term once, against the doubled operand d = 2a (limbs < 2^30).
// A column holds at most 4 cross terms of < 2^59, one square of < 2^58 and 9 reduction terms of < 2^58: < 2^62.4.
template <class C> ZK_DEV Fp29<C> sqr29(const Fp29<C>& a) {
    using L = Lim29<C>;
    Fp29<C> r;
    uint32_t m[9], d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) mad29(acc, d[i], a.l[k - i]);
        if ((k & 1) == 0) mad29(acc, a.l[k / 2], a.l[k / 2]);
#pragma unroll
        for (int i = 0; i < k; i++) mad29c(acc, m[i], L::p(k - i));
        m[k] = ((uint32_t)acc * L::NP) & M29;
        mad29c(acc, m[k], L::p(0));
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) mad29(acc, d[i], a.l[k - i]);
        if ((k & 1) == 0) mad29(acc, a.l[k / 2], a.l[k / 2]);
#pragma unroll
        for (int i = k - 8; i < 9; i++) mad29c(acc, m[i], L::p(k - i));
        r.l[k - 9] = (k == 17) ? (uint32_t)acc : ((uint32_t)acc & M29);
        acc >>= 29;
    }
    return r;
}
// carry propagation: limbs 0..7 back below 2^29 (input limbs < 2^32 - 2^3, value unchanged)
template <class C> ZK_DEV void norm29(Fp29<C>& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) { a.l[i + 1] += a.l[i] >> 29; a.l[i] &= M29; }
}
template <class C> ZK_DEV Fp29<C> add29(const Fp29<C>& a, const Fp29<C>& b) {      // limb-wise, NOT normalised
    Fp29<C> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// limb i of K*p in the redundant form  L_0 + 2^29, L_i + 2^29 - 1 (0 < i < 8), L_8 - 1:  same value, every limb below the top one >= 2^29 - 1
template <class C, int K> ZK_DEV constexpr uint32_t off29(int i) {
    // limbs of K*p by schoolbook on the 29-bit limbs of p
    uint64_t carry = 0;
    uint32_t li = 0;
    for (int j = 0; j <= i; j++) { uint64_t v = (uint64_t)Lim29<C>::p(j) * (uint64_t)K + carry; li = (j == 8) ? (uint32_t)v : ((uint32_t)v & M29); carry = v >> 29; }
    return i == 0 ? li + (1u << 29) : (i == 8 ? li - 1u : li + (1u << 29) - 1u);
}
// t + K p - b  (b normalised, value(b) < K p - 2^232); result NOT normalised: limbs grow by < 2^30 per call
template <class C, int K> ZK_DEV Fp29<C> sub29(const Fp29<C>& t, const Fp29<C>& b) {
    static_assert(K >= 1 && K <= 15, "offset multiple");
    Fp29<C> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = t.l[i] + (off29<C, K>(i) - b.l[i]);
    return r;
}
template <class C> ZK_DEV Fp29<C> zero29() { Fp29<C> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r; }
template <class C> ZK_DEV Fp29<C> one29() { Fp29<C> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = Lim29<C>::one(i);
    return r; }
// value == 0 mod p for a NORMALISED element below 2^258: the value is k p with k = v * p^-1 mod 2^29 (k < 16 then); everything else
// leaves after three instructions
template <class C> ZK_DEV bool is_zero29(const Fp29<C>& a) {
    const uint32_t k = (a.l[0] * Lim29<C>::PINV) & M29;
    if (k >= 32u) return false;
    uint64_t carry = 0;
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { uint64_t v = (uint64_t)Lim29<C>::p(i) * k + carry; uint32_t li = (i == 8) ? (uint32_t)v : ((uint32_t)v & M29); carry = v >> 29; diff |= li ^ a.l[i]; }
    return diff == 0;
}
// 8 packed words (canonical value) -> 9 limbs, no arithmetic
template <class C> ZK_DEV Fp29<C> unpack29(const uint32_t (&w)[8]) {
    Fp29<C> r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k, wi = bit >> 5, sh = bit & 31;
        uint64_t v = w[wi];
        if (wi + 1 < 8) v |= (uint64_t)w[wi + 1] << 32;
        r.l[k] = (k == 8) ? (uint32_t)(v >> sh) : ((uint32_t)(v >> sh) & M29);
    }
    return r;
}
template <class C> ZK_DEV Fp29<C> load29_packed(const uint32_t* p) {               // 32 bytes, 16-byte aligned
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return unpack29<C>(w);
}
// reference format (Montgomery factor 2^256, canonical, 8 words) -> R'-form limbs: one multiplication by 2^266 mod p
template <class C> ZK_DEV Fp29<C> from_r256(const uint32_t* p) {
    Fp29<C> k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = Lim29<C>::kin(i);
    return mul29(load29_packed<C>(p), k);
}
// R'-form limbs (any lazy value below 2^258) -> 8 canonical words: multiply by 2^256 mod p (the reference's R-form) or, KEEP29, by R' mod p
// (the value stays in R'-form: a later load is unpack29 alone); the product is < 1.6 p: subtract p while >= p, pack
template <class C, bool KEEP29 = false> ZK_DEV void store_r256(uint32_t* dst, const Fp29<C>& a_in) {
    Fp29<C> k, a = a_in;
    norm29(a);
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = KEEP29 ? Lim29<C>::one(i) : Lim29<C>::kout(i);
    Fp29<C> t = mul29(a, k);
#pragma unroll 1
    for (int rep = 0; rep < 2; rep++) {
        // d = t - p with signed carries; keep it if non-negative
        int32_t d[9], c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { int32_t v = (int32_t)t.l[i] - (int32_t)Lim29<C>::p(i) + c; if (i < 8) { c = v >> 29; d[i] = v & (int32_t)M29; } else d[i] = v; }
        const bool ge = d[8] >= 0;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = ge ? (uint32_t)d[i] : t.l[i];
    }
    uint32_t w[8];
    uint64_t acc = 0;
    int bits = 0, wi = 0;
#pragma unroll
    for (int kk = 0; kk < 9; kk++) {
        acc |= (uint64_t)t.l[kk] << bits; bits += 29;
        if (bits >= 32 && wi < 8) { w[wi++] = (uint32_t)acc; acc >>= 32; bits -= 32; }
    }
    if (wi < 8) w[wi] = (uint32_t)acc;
    uint4* q = reinterpret_cast<uint4*>(dst);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]); q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
template <class C> ZK_DEV void store_r29(uint32_t* dst, const Fp29<C>& a) { store_r256<C, true>(dst, a); }

}  // namespace zkmi
