// tools/field29_hosttest.hip — runs the __host__ __device__ arithmetic of csrc/field29.cuh and the point formulas of csrc/msm29.cuh ON THE CPU,
// driven over stdin/stdout by tests/test_field29_host.py, which checks every result against Python big integers. The build container has no
// GPU: this is how the unsaturated-limb code (9 x 29 bits, 14 x 28 bits) is verified before it is sent to one. Only the MAC differs between
// the two compilations (inline v_mad_u64_u32 on the device, a 64-bit multiply-add here); limb bounds, offsets, carries and formulas are shared.
//
// build: hipcc --offload-arch=gfx950 --cuda-host-only -O0 -std=c++17 -DZK29_CHECK [-DZK29_BOUNDS] -Isnarkjs_amd/csrc tools/field29_hosttest.hip -o tools/bin/field29_hosttest
// protocol: one request per line "<op> <curve> <hex words...>", one reply line of hex words (or "ERR ...").
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <sstream>
#include <iostream>
#include "msm29.cuh"
#include "ntt29.cuh"

using namespace zkmi;

template <class C> static Fp29<C> rd(const std::vector<uint32_t>& v, size_t& at) {
    Fp29<C> r;
    for (int i = 0; i < Lim29<C>::NL; i++) r.l[i] = v.at(at++);
    return r;
}
template <class C> static void wr(std::vector<uint32_t>& o, const Fp29<C>& a) { for (int i = 0; i < Lim29<C>::NL; i++) o.push_back(a.l[i]); }

// -DZK29_BOUNDS (field29.cuh): the harness states the CONTRACT of each function under test as worst-case bounds of its inputs and checks the bounds the
// arithmetic derives for the outputs against the invariants the function promises; without the flag these do nothing
#if defined(ZK29_SHADOW)
template <class C> static void bound_in(Fp29<C>& x, double bv) { x.bv = bv; x.bl = b29::lowmax<C>(); x.bt = -1.0; }      // normalised, value <= bv p
template <class C> static void bound_out(const Fp29<C>& x, double lim, const char* what) { b29::need(b29::normalised(x) && x.bv <= lim, what, x.bv, lim); }
#else
template <class C> static void bound_in(Fp29<C>&, double) {}
template <class C> static void bound_out(const Fp29<C>&, double, const char*) {}
#endif
// invariants of the accumulators (units of p; msm29.cuh): G1 XYZZ; G2 XYZZ (LDS-parked) of the reduction kernels; the same in the accumulation kernel, whose
// doubling branch leaves ZZ = V (a square: its c1 is a doubled product) and ZZZ = W as they come; G2 Jacobian (LDS-parked, 14-limb curve)
static const double INV_G1[4] = {7.3, 3.3, 1.1, 1.1}, INV_G2[4] = {8.4, 3.8, 1.1, 1.1}, INV_G2A[4] = {8.4, 3.8, 2.2, 1.2}, INV_G2J[3] = {8.4, 3.8, 1.1};
template <class C> static void bound_in(XYZZ29<C>& a) { bound_in(a.X, INV_G1[0]); bound_in(a.Y, INV_G1[1]); bound_in(a.ZZ, INV_G1[2]); bound_in(a.ZZZ, INV_G1[3]); }
template <class C> static void bound_out(const XYZZ29<C>& a) {
    bound_out(a.X, INV_G1[0], "G1 accumulator: X over its invariant"); bound_out(a.Y, INV_G1[1], "G1 accumulator: Y over its invariant");
    bound_out(a.ZZ, INV_G1[2], "G1 accumulator: ZZ over its invariant"); bound_out(a.ZZZ, INV_G1[3], "G1 accumulator: ZZZ over its invariant");
}
// a parked Fq2 accumulator: every coordinate read, given its invariant as bound, written back / read and checked
template <class C, class Acc> static void park_in(const Acc& A, int ncoord, const double* inv) {
    for (int k = 0; k < ncoord; k++) { F2x<C> v; A.get(k, v); bound_in(v.c0, inv[k]); bound_in(v.c1, inv[k]); A.put(k, v); }
}
template <class C, class Acc> static void park_out(const Acc& A, int ncoord, const double* inv) {
    for (int k = 0; k < ncoord; k++) { F2x<C> v; A.get(k, v); bound_out(v.c0, inv[k], "G2 accumulator: a coordinate over its invariant"); bound_out(v.c1, inv[k], "G2 accumulator: a coordinate over its invariant"); }
}

template <class C, int K> static Fp29<C> sub_k(const Fp29<C>& t, const Fp29<C>& b) { return sub29<C, K>(t, b); }
template <class C> static bool sub_any(int K, const Fp29<C>& t, const Fp29<C>& b, Fp29<C>& r) {
    switch (K) {
#define ZK_CASE(k) case k: r = sub_k<C, k>(t, b); return true;
        ZK_CASE(1) ZK_CASE(2) ZK_CASE(3) ZK_CASE(4) ZK_CASE(5) ZK_CASE(6) ZK_CASE(7) ZK_CASE(8) ZK_CASE(9) ZK_CASE(11) ZK_CASE(15) ZK_CASE(16) ZK_CASE(32) ZK_CASE(64)
#undef ZK_CASE
    }
    return false;
}

template <class C> static std::string run(const std::string& op, const std::vector<uint32_t>& v) {
    constexpr int N = C::N, NL = Lim29<C>::NL;
    std::vector<uint32_t> o;
    size_t at = 0;
    if (op == "mul") { auto a = rd<C>(v, at), b = rd<C>(v, at); wr(o, mul29(a, b)); }
    else if (op == "mul2") { auto a0 = rd<C>(v, at), b0 = rd<C>(v, at), a1 = rd<C>(v, at), b1 = rd<C>(v, at); wr(o, mul29_2(a0, b0, a1, b1)); }
    else if (op == "mul4") {
        Fp29<C> x[8];
        for (auto& e : x) e = rd<C>(v, at);
        wr(o, mul29_4(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]));
    }
    else if (op == "sqr") { auto a = rd<C>(v, at); wr(o, sqr29(a)); }
    else if (op == "norm") { auto a = rd<C>(v, at); norm29(a); wr(o, a); }
    else if (op == "sub") { int K = (int)v.at(at++); auto t = rd<C>(v, at), b = rd<C>(v, at); Fp29<C> r; if (!sub_any<C>(K, t, b, r)) return "ERR K"; wr(o, r); }
    else if (op == "iszero") { auto a = rd<C>(v, at); o.push_back(is_zero29(a) ? 1u : 0u); }
    else if (op == "canon") { auto a = rd<C>(v, at); canon29(a); wr(o, a); }
    else if (op == "unpack") { uint32_t w[N]; for (int i = 0; i < N; i++) w[i] = v.at(at++); wr(o, unpack29<C>(w)); }
    else if (op == "pack") { auto a = rd<C>(v, at); uint32_t w[N]; pack29<C>(w, a); for (int i = 0; i < N; i++) o.push_back(w[i]); }
    else if (op == "store" || op == "store29") {
        auto a = rd<C>(v, at);
        alignas(16) uint32_t w[N];
        if (op == "store") store_r256<C, false>(w, a); else store_r256<C, true>(w, a);
        for (int i = 0; i < N; i++) o.push_back(w[i]);
    }
    else if (op == "fromr") { alignas(16) uint32_t w[N]; for (int i = 0; i < N; i++) w[i] = v.at(at++); wr(o, from_r256<C>(w)); }
    else if (op == "madd" || op == "padd") {
        // in: inf, X, Y, ZZ, ZZZ, then the operand (x, y | X, Y, ZZ, ZZZ); out: inf, X, Y, ZZ, ZZZ
        bool inf = v.at(at++) != 0;
        XYZZ29<C> acc;
        acc.X = rd<C>(v, at); acc.Y = rd<C>(v, at); acc.ZZ = rd<C>(v, at); acc.ZZZ = rd<C>(v, at);
        bound_in(acc);
        if (op == "madd") { Aff29<C> q; q.x = rd<C>(v, at); q.y = rd<C>(v, at); bound_in(q.x, 1.0); bound_in(q.y, 2.0); madd29(acc, inf, q); }
        else { XYZZ29<C> p; p.X = rd<C>(v, at); p.Y = rd<C>(v, at); p.ZZ = rd<C>(v, at); p.ZZZ = rd<C>(v, at); bound_in(p); padd29(acc, inf, p); }
        if (!inf) bound_out(acc);
        o.push_back(inf ? 1u : 0u);
        wr(o, acc.X); wr(o, acc.Y); wr(o, acc.ZZ); wr(o, acc.ZZZ);
    }
    else if (op == "storept" || op == "storept29") {
        XYZZ29<C> a;
        bool inf = v.at(at++) != 0;
        a.X = rd<C>(v, at); a.Y = rd<C>(v, at); a.ZZ = rd<C>(v, at); a.ZZZ = rd<C>(v, at);
        alignas(16) uint32_t w[4 * N];
        if (op == "storept") store_xyzz29<C, false>(w, a, inf); else store_xyzz29<C, true>(w, a, inf);
        for (int i = 0; i < 4 * N; i++) o.push_back(w[i]);
        if (op == "storept29") {                 // and back through load_xyzz29
            XYZZ29<C> b;
            const bool ok = load_xyzz29(b, w);
            o.push_back(ok ? 1u : 0u);
            if (ok) { wr(o, b.X); wr(o, b.Y); wr(o, b.ZZ); wr(o, b.ZZZ); }
        }
    }
    else if (op == "madd2seq") {
        // G2: a sequence of mixed additions into one LDS-parked accumulator (T = 1: the "LDS" is a host array), packed or not as the kernel
        // has it for this curve. in: count, then count x (neg, x.c0, x.c1, y.c0, y.c1) canonical R'-form limbs; out: inf + 8 N words (store_xyzz29_lds)
        typedef LdsAcc29<C, 1, Accum29G2<C>::PACK> Acc;
        std::vector<uint32_t> lds(Accum29G2<C>::lds_bytes / 4 / Accum29G2<C>::T, 0xdeadbeefu);      // exactly what the kernel gives a lane
        const Acc A{lds.data()};
        bool inf = true;
        const uint32_t cnt = v.at(at++);
        for (uint32_t k = 0; k < cnt; k++) {
            const bool neg = v.at(at++) != 0;
            F2x<C> qx, qy;
            qx.c0 = rd<C>(v, at); qx.c1 = rd<C>(v, at); qy.c0 = rd<C>(v, at); qy.c1 = rd<C>(v, at);
            if (neg) { qy.c0 = neg29<C, 2>(qy.c0); qy.c1 = neg29<C, 2>(qy.c1); }
            constexpr int NCO = Accum29G2<C>::JAC ? 3 : 4;
            const double* inv = Accum29G2<C>::JAC ? INV_G2J : INV_G2A;
            bound_in(qx.c0, 1.0); bound_in(qx.c1, 1.0); bound_in(qy.c0, 2.0); bound_in(qy.c1, 2.0);
            if (!inf) park_in<C>(A, NCO, inv);
            Accum29G2<C>::madd(A, inf, qx, qy, [&](F2x<C>& x2, F2x<C>& y2) { x2 = qx; y2 = qy; });
            if (!inf) park_out<C>(A, NCO, inv);
        }
        alignas(16) uint32_t w[8 * N];
        if (!inf) park_in<C>(A, Accum29G2<C>::JAC ? 3 : 4, Accum29G2<C>::JAC ? INV_G2J : INV_G2A);
        Accum29G2<C>::template store<false>(w, A, inf);
        o.push_back(inf ? 1u : 0u);
        for (int i = 0; i < 8 * N; i++) o.push_back(w[i]);
    }
    else if (op == "madd2xyzz") {
        // r06: the XYZZ form of the G2 accumulation for EVERY curve, accumulator unpacked — the arithmetic of k_msm_accum29_g2s (one Fq2 component per lane:
        // madd29_split forms, per component, exactly the products, offsets and carry passes of madd29_lds; the 14-limb curve's default was the packed
        // Jacobian before, so its XYZZ offsets had never met the worst-case check). Same in / out as madd2seq.
        typedef LdsAcc29<C, 1, false> Acc;
        std::vector<uint32_t> lds((size_t)8 * Lim29<C>::NL, 0xdeadbeefu);
        const Acc A{lds.data()};
        bool inf = true;
        const uint32_t cnt = v.at(at++);
        for (uint32_t k = 0; k < cnt; k++) {
            const bool neg = v.at(at++) != 0;
            F2x<C> qx, qy;
            qx.c0 = rd<C>(v, at); qx.c1 = rd<C>(v, at); qy.c0 = rd<C>(v, at); qy.c1 = rd<C>(v, at);
            if (neg) { qy.c0 = neg29<C, 2>(qy.c0); qy.c1 = neg29<C, 2>(qy.c1); }
            bound_in(qx.c0, 1.0); bound_in(qx.c1, 1.0); bound_in(qy.c0, 2.0); bound_in(qy.c1, 2.0);
            if (!inf) park_in<C>(A, 4, INV_G2A);
            madd29_lds<C>(A, inf, qx, qy);
            if (!inf) park_out<C>(A, 4, INV_G2A);
        }
        alignas(16) uint32_t w[8 * N];
        if (!inf) park_in<C>(A, 4, INV_G2A);
        store_xyzz29_lds<C, Acc, false>(w, A, inf);
        o.push_back(inf ? 1u : 0u);
        for (int i = 0; i < 8 * N; i++) o.push_back(w[i]);
    }
    else if (op == "padd2") {
        // G2 bucket reduction: groups of signed affine points are accumulated into buckets the way the accumulation kernel does and stored as
        // R'-form words; a reduction accumulator (XYZZ, LDS-parked, the reduction kernel's packing) then folds the buckets from the words
        // (padd29_lds), and a second one takes the first one twice through the accumulator-to-accumulator form (doubling branch).
        // in: groups, per group: count, count x (neg, x.c0, x.c1, y.c0, y.c1); out: inf + 8 N R-form words of the sum, the same of its double
        typedef LdsAcc29<C, 1, Accum29G2<C>::PACK> AccA;
        typedef LdsAcc29<C, 1, Reduce29G2<C>::PACK> AccR;
        std::vector<uint32_t> ldsA(Accum29G2<C>::lds_bytes / 4 / Accum29G2<C>::T), ldsR(Reduce29G2<C>::lds_bytes / 4 / Reduce29G2<C>::T, 0xdeadbeefu), ldsD(ldsR.size(), 0xdeadbeefu);
        const AccR Rr{ldsR.data()}, Dd{ldsD.data()};
        bool rinf = true, dinf = true;
        const uint32_t groups = v.at(at++);
        for (uint32_t gi = 0; gi < groups; gi++) {
            const AccA A{ldsA.data()};
            bool inf = true;
            const uint32_t cnt = v.at(at++);
            for (uint32_t k = 0; k < cnt; k++) {
                const bool neg = v.at(at++) != 0;
                F2x<C> qx, qy;
                qx.c0 = rd<C>(v, at); qx.c1 = rd<C>(v, at); qy.c0 = rd<C>(v, at); qy.c1 = rd<C>(v, at);
                if (neg) { qy.c0 = neg29<C, 2>(qy.c0); qy.c1 = neg29<C, 2>(qy.c1); }
                bound_in(qx.c0, 1.0); bound_in(qx.c1, 1.0); bound_in(qy.c0, 2.0); bound_in(qy.c1, 2.0);
                if (!inf) park_in<C>(A, Accum29G2<C>::JAC ? 3 : 4, Accum29G2<C>::JAC ? INV_G2J : INV_G2A);
                Accum29G2<C>::madd(A, inf, qx, qy, [&](F2x<C>& x2, F2x<C>& y2) { x2 = qx; y2 = qy; });
            }
            alignas(16) uint32_t w[8 * N];
            if (!inf) park_in<C>(A, Accum29G2<C>::JAC ? 3 : 4, Accum29G2<C>::JAC ? INV_G2J : INV_G2A);
            Accum29G2<C>::template store<true>(w, A, inf);
            if (!xyzz29_words_inf_g2<C>(w) != !inf) return "ERR infinity encoding";
            if (!inf) {
                if (!rinf) park_in<C>(Rr, 4, INV_G2);
                padd29_lds<C>(Rr, rinf, [&](int k, F2x<C>& x) { x.c0 = load29_packed<C>(w + k * 2 * N); x.c1 = load29_packed<C>(w + k * 2 * N + N); });
                if (!rinf) park_out<C>(Rr, 4, INV_G2);
            }
        }
        for (int rep = 0; rep < 2; rep++)
            if (!rinf) {
                park_in<C>(Rr, 4, INV_G2);                                   // the operand: any accumulator within the invariants
                if (!dinf) park_in<C>(Dd, 4, INV_G2);
                padd29_lds<C>(Dd, dinf, [&](int k, F2x<C>& x) { Rr.get(k, x); });
                if (!dinf) park_out<C>(Dd, 4, INV_G2);
            }
        if (!rinf) park_in<C>(Rr, 4, INV_G2);
        if (!dinf) park_in<C>(Dd, 4, INV_G2);
        alignas(16) uint32_t w[8 * N];
        store_xyzz29_lds<C, AccR, false>(w, Rr, rinf);
        o.push_back(rinf ? 1u : 0u);
        for (int i = 0; i < 8 * N; i++) o.push_back(w[i]);
        store_xyzz29_lds<C, AccR, false>(w, Dd, dinf);
        o.push_back(dinf ? 1u : 0u);
        for (int i = 0; i < 8 * N; i++) o.push_back(w[i]);
    }
    else if (op == "rowsum") {
        // the reduction kernel's flow for one wave on the host: 64 lanes with accumulators at stride T = 64 in one "LDS" array; lane l places
        // bucket l (R'-form words; all-zero words = empty), then the six tree levels. in: 64 x 8 N words; out: inf + 8 N R-form words
        constexpr int T = 64;
        typedef LdsAcc29<C, T, Reduce29G2<C>::PACK> Acc;
        std::vector<uint32_t> lds((size_t)T * 8 * Acc::EW, 0xdeadbeefu);
        std::vector<uint32_t> words(v.begin() + at, v.begin() + at + 64 * 8 * N);
        at += 64 * 8 * N;
        bool inf[T];
        for (int l = 0; l < T; l++) {
            const Acc A{lds.data() + l};
            inf[l] = true;
            alignas(16) uint32_t w[8 * N];
            for (int i = 0; i < 8 * N; i++) w[i] = words[(size_t)l * 8 * N + i];
            if (!xyzz29_words_inf_g2<C>(w)) padd29_lds<C>(A, inf[l], [&](int k, F2x<C>& x) { x.c0 = load29_packed<C>(w + k * 2 * N); x.c1 = load29_packed<C>(w + k * 2 * N + N); });
        }
        for (int d = 1; d < 64; d <<= 1)
            for (int l = 0; l < T; l++)
                if ((l & (2 * d - 1)) == 0 && !inf[l + d]) {
                    const Acc A{lds.data() + l}, Pn{lds.data() + l + d};
                    park_in<C>(Pn, 4, INV_G2);
                    if (!inf[l]) park_in<C>(A, 4, INV_G2);
                    padd29_lds<C>(A, inf[l], [&](int k, F2x<C>& x) { Pn.get(k, x); });
                    if (!inf[l]) park_out<C>(A, 4, INV_G2);
                }
        alignas(16) uint32_t w[8 * N];
        const Acc A0{lds.data()};
        if (!inf[0]) park_in<C>(A0, 4, INV_G2);
        store_xyzz29_lds<C, Acc, false>(w, A0, inf[0]);
        o.push_back(inf[0] ? 1u : 0u);
        for (int i = 0; i < 8 * N; i++) o.push_back(w[i]);
    }
    else if (op == "reduce") {
        if constexpr (Lim29<C>::NL == 9) { auto a = rd<C>(v, at); reduce29_small(a); wr(o, a); } else return "ERR form";
    }
    else if (op == "bfly") {                 // in: x, y, w, has_w; out: x', y' (ntt29.cuh: one DIT butterfly as the tile stages run it)
        if constexpr (Lim29<C>::NL == 9) {
            auto x = rd<C>(v, at), y = rd<C>(v, at), w = rd<C>(v, at);
            const bool has_w = v.at(at++) != 0;
            // contract of a tile stage (ntt29.cuh): x and y lazy values of earlier stages (at most 1.3 r + 2 r per stage before them: 17.3 r ahead of the
            // ninth and last), the twiddle canonical; without a twiddle (first stage) y is a fresh product or canonical (1.3 r)
            bound_in(x, 17.3); bound_in(y, has_w ? 17.3 : 1.3); bound_in(w, 1.0);
            if (has_w) y = mul29(y, w);
            ntt29_bfly(x, y, Fp29<C>(y));
            bound_out(x, 19.3, "NTT butterfly: x over 19.3 r"); bound_out(y, 19.3, "NTT butterfly: y over 19.3 r");
            { Fp29<C> z = x; reduce29_small(z); }          // what the last pass does with it: the range of the final reduction
            wr(o, x); wr(o, y);
        } else return "ERR form";
    }
    else if (op == "nttin") {               // in: x (a record of an earlier pass), t_lo, t_hi, rowinc, scale (R'-form constants); out: the tile element a pass starts from
        if constexpr (Lim29<C>::NL == 9) {
            auto x = rd<C>(v, at), tlo = rd<C>(v, at), thi = rd<C>(v, at), inc = rd<C>(v, at), sc = rd<C>(v, at);
            // contract (ntt29.cuh: k_ntt29_pass_strided / _last): x any lazy value a pass can leave (19.3 r), the table entries canonical
            bound_in(x, 19.3); bound_in(tlo, 1.0); bound_in(thi, 1.0); bound_in(inc, 1.0); bound_in(sc, 1.0);
            x = mul29(x, mul29(tlo, thi));                 // ntt29_pow
            x = mul29(x, inc);
            x = mul29(x, sc);
            bound_out(x, 1.3, "NTT tile element: over the 1.3 r the first stage assumes");
            wr(o, x);
        } else return "ERR form";
    }
    else if (op == "bounds") {              // violations recorded so far by the worst-case tracking (-DZK29_BOUNDS builds), log2 of the largest column x 1000
#if defined(ZK29_SHADOW)
        o.push_back((uint32_t)b29::failures()); o.push_back((uint32_t)(b29::max_column() > 0 ? log2(b29::max_column()) * 1000.0 : 0.0)); o.push_back(1u);
#else
        o.push_back(0u); o.push_back(0u); o.push_back(0u);
#endif
    }
    else if (op == "consts") {
        o.push_back(NL); o.push_back(Lim29<C>::B); o.push_back(N); o.push_back(Lim29<C>::NP); o.push_back(Lim29<C>::PINV);
        for (int i = 0; i < NL; i++) o.push_back(Lim29<C>::p(i));
        for (int i = 0; i < NL; i++) o.push_back(Lim29<C>::one(i));
        for (int i = 0; i < NL; i++) o.push_back(Lim29<C>::kin(i));
        for (int i = 0; i < NL; i++) o.push_back(Lim29<C>::kout(i));
    }
    else return "ERR op";
    if (at != v.size()) return "ERR trailing input";
    std::ostringstream ss;
    for (size_t i = 0; i < o.size(); i++) { if (i) ss << ' '; ss << std::hex << o[i]; }
    return ss.str();
}

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string op, curve, tok;
        is >> op >> curve;
        std::vector<uint32_t> v;
        while (is >> tok) v.push_back((uint32_t)strtoul(tok.c_str(), nullptr, 16));
        std::string out;
        try {
            if (curve == "bn254fq") out = run<Bn254Fq>(op, v);
            else if (curve == "bls12381fq") out = run<Bls12381Fq>(op, v);
            else if (curve == "bls12381fq_compact") out = run<Compact<Bls12381Fq>>(op, v);
            else if (curve == "bn254fr") out = run<Bn254Fr>(op, v);
            else if (curve == "bls12381fr") out = run<Bls12381Fr>(op, v);
            else out = "ERR curve";
        } catch (const std::exception& e) { out = std::string("ERR ") + e.what(); }
        std::cout << out << "\n" << std::flush;
    }
    return 0;
}
