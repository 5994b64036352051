// tools/plonk29_hosttest.hip — PLONK's quotient numerator on 29-bit limbs (csrc/plonk29.cuh) run ON THE CPU, two ways (tests/test_plonk29_host.py):
//   plonk29_hosttest run <curve> <in.bin> <out.bin>   the arithmetic: the three parts over every point of a small extended domain; the test compares
//                                                      t / tz with the oracle's literal MulZ expansion (oracle/plonk_oracle.py: mul2, mul4), bit for bit.
//                                                      Built with -DZK29_CHECK: the subtrahend-under-offset precondition of sub29 is checked exactly per value.
//   plonk29_hosttest bounds <curve>                    the SAME body instantiated with an interval type: every element carries (value bound in units
//                                                      of r, limb bound, normalised?) and every precondition of field29.cuh is checked for the WORST case —
//                                                      operand limbs of the products, column sums below 2^64, offsets, ranges of the final reductions.
// The build container has no GPU; only the multiply-add differs between this compilation and the device's.
// build: hipcc --offload-arch=gfx950 --cuda-host-only -O0 -std=c++17 -DZK29_CHECK [-DZK29_BOUNDS] -Isnarkjs_amd/csrc tools/plonk29_hosttest.hip -o tools/bin/plonk29_hosttest
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "plonk29.cuh"

namespace zkmi {

// ---- the interval model ---------------------------------------------------------------------------------------------------------------------------
static int g_fail = 0;
static double g_max_col = 0, g_max_top = 0, g_max_val = 0;
static void need(bool ok, const char* what, double got, double lim) {
    if (!ok) { g_fail++; fprintf(stdout, "FAIL %s: %.6g (limit %.6g)\n", what, got, lim); }
}
template <class C> struct Bnd29 {
    typedef C Cfg;
    double val = 0;      // value < val * r
    double limb = 0;     // limbs 0 .. NL-2 <= limb * (2^B - 1)
    bool norm = true;    // limbs 0 .. NL-2 < 2^B
};
template <class C> struct E29<Bnd29<C>> {
    typedef Bnd29<C> E;
    typedef Lim29<C> L;
    static double p_val() { double v = 0; for (int i = L::NL - 1; i >= 0; i--) v = v * ldexp(1.0, L::B) + (double)L::p(i); return v; }
    static double ratio() { return ldexp(1.0, L::B * L::NL) / p_val(); }                 // R' / r
    static double ptop() { return p_val() / ldexp(1.0, L::B * (L::NL - 1)); }            // r in units of the top limb
    static double top(const E& a) { return a.val * ptop() + 1; }                         // bound of the top limb
    static double lim(const E& a) { return fmax(a.limb * (ldexp(1.0, L::B) - 1), top(a)); }
    static E mk(double val, double limb, bool norm) { E e; e.val = val; e.limb = limb; e.norm = norm; if (val > g_max_val) g_max_val = val; need(top(e) < 4294967296.0, "top limb", top(e), 4294967296.0); if (top(e) > g_max_top) g_max_top = top(e); return e; }
    static E load(const uint32_t*) { return mk(1, 1, true); }
    static E load_shl(const uint32_t*) { return mk(ldexp(1.0, r29_shift<C>()), 1, true); }
    static E zero() { return mk(0, 0, true); }
    static E one() { return mk(1, 1, true); }
    // sum of products with one reduction: the column bound of field29.cuh's product scanning for the worst operands, the value of the result
    static E mulN(int n, const E* const* a, const E* const* b) {
        double col = 0, v = 0;
        int lazy = 0;
        for (int j = 0; j < n; j++) {
            col += L::NL * lim(*a[j]) * lim(*b[j]);
            v += a[j]->val * b[j]->val;
            lazy += !a[j]->norm; lazy += !b[j]->norm;
        }
        col += L::NL * ldexp(1.0, 2 * L::B) + ldexp(1.0, 36);
        need(col < 18446744073709551616.0, "column sum", col, 18446744073709551616.0);
        if (col > g_max_col) g_max_col = col;
        // field29.cuh's own wording of the operand conditions (stricter than the column sum): mul29: one operand < 2^(B+2) against a normalised one, or
        // both < 2^(B+1); mul29_2: one operand < 2^(B+1), the rest normalised; mul29_3 / mul29_4: all normalised
        if (n == 1) need((a[0]->norm && lim(*b[0]) < ldexp(1.0, L::B + 2)) || (b[0]->norm && lim(*a[0]) < ldexp(1.0, L::B + 2)) || (lim(*a[0]) < ldexp(1.0, L::B + 1) && lim(*b[0]) < ldexp(1.0, L::B + 1)), "mul29 operands", lim(*a[0]), lim(*b[0]));
        else if (n == 2) need(lazy <= 1, "mul29_2: more than one lazy operand", lazy, 1);
        else need(lazy == 0, "mul29_3/4: lazy operand", lazy, 0);
        for (int j = 0; j < n; j++) { need(lim(*a[j]) < ldexp(1.0, L::B + 2), "operand limb", lim(*a[j]), ldexp(1.0, L::B + 2)); need(lim(*b[j]) < ldexp(1.0, L::B + 2), "operand limb", lim(*b[j]), ldexp(1.0, L::B + 2)); }
        return mk(v / ratio() + 1, 1, true);
    }
    static E mul(const E& a, const E& b) { const E* aa[1] = {&a}; const E* bb[1] = {&b}; return mulN(1, aa, bb); }
    static E mul2(const E& a0, const E& b0, const E& a1, const E& b1) { const E* aa[2] = {&a0, &a1}; const E* bb[2] = {&b0, &b1}; return mulN(2, aa, bb); }
    static E mul3(const E& a0, const E& b0, const E& a1, const E& b1, const E& a2, const E& b2) { const E* aa[3] = {&a0, &a1, &a2}; const E* bb[3] = {&b0, &b1, &b2}; return mulN(3, aa, bb); }
    static E mul4(const E& a0, const E& b0, const E& a1, const E& b1, const E& a2, const E& b2, const E& a3, const E& b3) {
        const E* aa[4] = {&a0, &a1, &a2, &a3}; const E* bb[4] = {&b0, &b1, &b2, &b3}; return mulN(4, aa, bb);
    }
    static E add(const E& a, const E& b) {
        E r = mk(a.val + b.val, a.limb + b.limb, false);
        need(r.limb * (ldexp(1.0, L::B) - 1) < 4294967296.0, "add: limb", r.limb, 7);
        return r;
    }
    static E sub2(const E& t, const E& b) {
        need(b.norm, "sub29: subtrahend not normalised", 0, 0);
        need(b.val <= 2 - 1 / ptop(), "sub29<2>: subtrahend over the offset", b.val, 2 - 1 / ptop());
        E r = mk(t.val + 2, t.limb + 2.0 + 2.0 / (ldexp(1.0, L::B) - 1), false);
        need(r.limb * (ldexp(1.0, L::B) - 1) < 4294967296.0, "sub: limb", r.limb, 7);
        return r;
    }
    static void norm(E& a) {
        need(a.limb * (ldexp(1.0, L::B) - 1) < 4294967296.0 - ldexp(1.0, 32 - L::B), "norm29: limb", a.limb, 7);
        a = mk(a.val, 1, true);            // the top limb takes the carries: its bound follows from the value
    }
    static E reduce_lt32(E v) { norm(v); need(v.val <= 32, "reduce29_small: value", v.val, 32); need(top(v) < ldexp(1.0, 31), "reduce29_small: top limb", top(v), ldexp(1.0, 31)); return mk(1, 1, true); }
    static void store_lt3(uint32_t*, E v) { norm(v); need(v.val <= 3, "canon29: value", v.val, 3); }
    static void store_lt32(uint32_t*, E v) { reduce_lt32(v); }
};

template <class C> static int run_bounds() {
    PlonkTArgs g;
    memset(&g, 0, sizeof g);
    g.domain = 4; g.n_public = 3;
    PowTab w4{nullptr, nullptr, 2};
    plonk_t29_point<Bnd29<C>, 0>(g, w4, 5);
    plonk_t29_point<Bnd29<C>, 1>(g, w4, 5);
    plonk_t29_point<Bnd29<C>, 2>(g, w4, 5);
    printf("ratio %.3f max_value %.3f r  max_column 2^%.3f  max_top_limb 2^%.3f\n", E29<Bnd29<C>>::ratio(), g_max_val, log2(g_max_col), log2(g_max_top));
    printf(g_fail ? "FAILED %d\n" : "OK\n", g_fail);
    return g_fail ? 1 : 0;
}

// ---- the arithmetic ---------------------------------------------------------------------------------------------------------------------------------
// in.bin (uint32 words): domain, n_public, lb, then: a b c z qm ql qr qo qc s1 s2 s3 (4n elements each), lagrange (n_public * 5n), pub_a (n_public),
// k (PK_COUNT), k29 (PK_COUNT), pow lo (2^lb), pow hi (4n >> lb); out.bin: t, tz (4n elements each)
template <class C> static int run_points(const char* in_path, const char* out_path) {
    FILE* f = fopen(in_path, "rb");
    if (!f) { perror(in_path); return 2; }
    std::vector<uint32_t> buf;
    uint32_t w;
    while (fread(&w, 4, 1, f) == 1) buf.push_back(w);
    fclose(f);
    size_t at = 0;
    const uint32_t dom = buf.at(at++), npub = buf.at(at++), lb = buf.at(at++), n4 = 4 * dom;
    // 16-byte aligned copies (the loads are vector loads)
    std::vector<std::vector<uint4>> keep;
    auto take = [&](size_t elems) -> const uint32_t* {
        keep.emplace_back(elems * 2 + 1);
        uint32_t* d = reinterpret_cast<uint32_t*>(keep.back().data());
        for (size_t i = 0; i < elems * 8; i++) d[i] = buf.at(at++);
        return d;
    };
    PlonkTArgs g;
    memset(&g, 0, sizeof g);
    g.domain = dom; g.n_public = npub;
    g.a = take(n4); g.b = take(n4); g.c = take(n4); g.z = take(n4); g.qm = take(n4); g.ql = take(n4); g.qr = take(n4); g.qo = take(n4); g.qc = take(n4);
    g.s1 = take(n4); g.s2 = take(n4); g.s3 = take(n4);
    g.lagrange = take((size_t)npub * 5 * dom); g.pub_a = take(npub);
    g.k = take(PK_COUNT); g.k29 = take(PK_COUNT);
    PowTab w4;
    w4.lb = lb; w4.lo = take((size_t)1 << lb); w4.hi = take(n4 >> lb);
    if (at != buf.size()) { fprintf(stderr, "input size: %zu words read of %zu\n", at, buf.size()); return 2; }
    std::vector<uint4> t(n4 * 2), tz(n4 * 2);
    g.t = reinterpret_cast<uint32_t*>(t.data()); g.tz = reinterpret_cast<uint32_t*>(tz.data());
    for (uint32_t i = 0; i < n4; i++) plonk_t29_point<Fp29<C>, 0>(g, w4, i);
    for (uint32_t i = 0; i < n4; i++) plonk_t29_point<Fp29<C>, 1>(g, w4, i);
    for (uint32_t i = 0; i < n4; i++) plonk_t29_point<Fp29<C>, 2>(g, w4, i);
    f = fopen(out_path, "wb");
    if (!f) { perror(out_path); return 2; }
    fwrite(g.t, 32, n4, f); fwrite(g.tz, 32, n4, f);
    fclose(f);
#if defined(ZK29_SHADOW)
    // -DZK29_BOUNDS: the arithmetic itself carried worst-case bounds through every primitive (field29.cuh): a second, independent check of what `bounds` models
    if (b29::failures()) { fprintf(stderr, "%d precondition(s) violated in the worst case\n", b29::failures()); return 3; }
#endif
    return 0;
}

}  // namespace zkmi

using namespace zkmi;

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: plonk29_hosttest run|bounds <bn254fr|bls12381fr> [in.bin out.bin]\n"); return 2; }
    const std::string mode = argv[1], curve = argv[2];
    if (mode == "bounds") return curve == "bn254fr" ? run_bounds<Bn254Fr>() : run_bounds<Bls12381Fr>();
    if (mode == "run" && argc >= 5) return curve == "bn254fr" ? run_points<Bn254Fr>(argv[3], argv[4]) : run_points<Bls12381Fr>(argv[3], argv[4]);
    return 2;
}
