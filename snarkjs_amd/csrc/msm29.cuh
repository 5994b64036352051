// snarkjs_amd/csrc/msm29.cuh — bucket accumulation over resident window tables on unsaturated 29-bit limbs (field29.cuh).
//
// Same job as k_msm_accum (msm.cuh: one lane per bucket or bucket share, XYZZ mixed additions over the sorted digit lists), but the
// field arithmetic of the hot loop runs on 9 x 29-bit limbs with lazy additions: 10 products of ~207 instructions instead of ~290 per
// mixed addition. Boundaries keep the library's formats: window tables are canonical 8-word values in R'-form (k_table_to_r29 converts a
// table once, when it is built), buckets and lane partials are written in the reference's R-form (store_r256), so the lane-partial
// trees and the bucket reduction are unchanged.
//
// Value bounds (units of p; R'/p = 170): a product of a and b comes back below a*b/170 + 1. Offsets K of the lazy subtractions are
// chosen from the bounds noted on each line of madd29.
#pragma once
#include "field29.cuh"
#include "msm.cuh"

namespace zkmi {

template <class C> struct Aff29 { Fp29<C> x, y; };
template <class C> struct XYZZ29 { Fp29<C> X, Y, ZZ, ZZZ; };          // invariants: X <= 7.3, Y <= 3.3, ZZ, ZZZ <= 1.1; all normalised

// doubling of an affine point (mdbl-2008-s-1, a = 0): the rare equal-points branch of madd29
template <class C> ZK_DEV void dbl_affine29(XYZZ29<C>& r, const Aff29<C>& q) {
    Fp29<C> U = add29(q.y, q.y); norm29(U);                                     // <= 4
    const Fp29<C> V = mul29(U, U), W = mul29(U, V), S = mul29(q.x, V), xx = mul29(q.x, q.x);       // <= 1.1
    Fp29<C> M = add29(add29(xx, xx), xx); norm29(M);                            // <= 3.3
    Fp29<C> X3 = sub29<C, 2>(sub29<C, 2>(mul29(M, M), S), S); norm29(X3);       // <= 5.1
    const Fp29<C> T = sub29<C, 6>(S, X3);                                       // <= 7.1, limbs < 2^31
    Fp29<C> Y3 = sub29<C, 2>(mul29(M, T), mul29(W, q.y)); norm29(Y3);           // <= 3.2
    r.X = X3; r.Y = Y3; r.ZZ = V; r.ZZZ = W;
}
// acc += q (q affine, not the point at infinity, x canonical, y <= 2 normalised); inf = accumulator is the point at infinity.
// 8 products + 2 squarings (sqr29) with 9 reductions: the last two products share one (mul29_2).
template <class C> ZK_DEV void madd29(XYZZ29<C>& acc, bool& inf, const Aff29<C>& q) {
    if (inf) { acc.X = q.x; acc.Y = q.y; acc.ZZ = one29<C>(); acc.ZZZ = one29<C>(); inf = false; return; }
    const Fp29<C> U2 = mul29(q.x, acc.ZZ), S2 = mul29(q.y, acc.ZZZ);           // <= 1.1
    Fp29<C> P = sub29<C, 8>(U2, acc.X); norm29(P);                              // X <= 7.3 < 8;  P <= 9.1
    Fp29<C> R = sub29<C, 4>(S2, acc.Y); norm29(R);                              // Y <= 3.3 < 4;  R <= 5.1
    if (is_zero29(P)) {
        if (is_zero29(R)) dbl_affine29(acc, q); else inf = true;
        return;
    }
    const Fp29<C> PP = sqr29(P);                                                // <= 1.49
    const Fp29<C> PPP = mul29(P, PP), Q = mul29(acc.X, PP);                     // <= 1.08, 1.07
    Fp29<C> X3 = sub29<C, 2>(sub29<C, 2>(sub29<C, 2>(sqr29(R), PPP), Q), Q); norm29(X3);          // <= 1.16 + 6 = 7.16
    Fp29<C> T = sub29<C, 8>(Q, X3); norm29(T);                                  // <= 9.1
    // Y3 = R T - Y1 PPP as ONE double product with one reduction: (R T + (4p - Y1) PPP) / R' + p <= (46.4 + 4.4) / 169 + 1 = 1.3
    const Fp29<C> Y3 = mul29_2(R, T, sub29<C, 4>(zero29<C>(), acc.Y), PPP);     // 4p - Y1: limbs < 2^30 (not normalised), as in f2mul
    acc.ZZ = mul29(acc.ZZ, PP); acc.ZZZ = mul29(acc.ZZZ, PPP);
    acc.X = X3; acc.Y = Y3;
}
// KEEP29: canonical words in R'-form (buckets that the 29-bit row/column sums read back with shifts alone); else the reference's R-form
template <class C, bool KEEP29 = false> ZK_DEV void store_xyzz29(uint32_t* dst, const XYZZ29<C>& a, bool inf) {
    if (inf) {
#pragma unroll
        for (int i = 0; i < 8; i++) reinterpret_cast<uint4*>(dst)[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    store_r256<C, KEEP29>(dst, a.X); store_r256<C, KEEP29>(dst + 8, a.Y); store_r256<C, KEEP29>(dst + 16, a.ZZ); store_r256<C, KEEP29>(dst + 24, a.ZZZ);
}
// a point stored by store_xyzz29<C, true> (all-zero = infinity)
template <class C> ZK_DEV bool load_xyzz29(XYZZ29<C>& a, const uint32_t* src) {
    const uint4* q = reinterpret_cast<const uint4*>(src);
    const uint4 z0 = q[4], z1 = q[5];
    if (!(z0.x | z0.y | z0.z | z0.w | z1.x | z1.y | z1.z | z1.w)) return false;
    a.X = load29_packed<C>(src); a.Y = load29_packed<C>(src + 8); a.ZZ = load29_packed<C>(src + 16); a.ZZZ = load29_packed<C>(src + 24);
    return true;
}
// acc = 2 acc for a general XYZZ accumulator (dbl-2008-s-1, a = 0): the rare equal-points branch of padd29. Same invariants as madd29.
template <class C> ZK_DEV void dbl_xyzz29(XYZZ29<C>& r) {
    Fp29<C> U = add29(r.Y, r.Y); norm29(U);                                     // <= 6.6
    const Fp29<C> V = sqr29(U), W = mul29(U, V), S = mul29(r.X, V), xx = sqr29(r.X);            // <= 1.26, 1.05, 1.06, 1.32
    Fp29<C> M = add29(add29(xx, xx), xx); norm29(M);                            // <= 3.96
    Fp29<C> X3 = sub29<C, 2>(sub29<C, 2>(sqr29(M), S), S); norm29(X3);          // <= 1.1 + 4 = 5.1
    Fp29<C> T = sub29<C, 6>(S, X3); norm29(T);                                  // <= 7.1
    const Fp29<C> Y3 = mul29_2(M, T, sub29<C, 2>(zero29<C>(), W), r.Y);         // (M T + (2p - W) Y) / R' + p <= (28.1 + 6.6) / 169 + 1 = 1.21
    r.ZZ = mul29(V, r.ZZ); r.ZZZ = mul29(W, r.ZZZ);
    r.X = X3; r.Y = Y3;
}
// acc += p for two general XYZZ points (add-2008-s, 12M + 2S, the last two products share one reduction). Both operands within the
// invariants of madd29 (X <= 7.3, Y <= 3.3, ZZ, ZZZ <= 1.1, normalised; a point just unpacked from memory is canonical); the result too.
template <class C> ZK_DEV void padd29(XYZZ29<C>& acc, bool& inf, const XYZZ29<C>& p) {
    if (inf) { acc = p; inf = false; return; }
    const Fp29<C> U1 = mul29(acc.X, p.ZZ), U2 = mul29(p.X, acc.ZZ);             // <= 1.05
    Fp29<C> P = sub29<C, 2>(U2, U1); norm29(P);                                 // <= 3.05
    const Fp29<C> S1 = mul29(acc.Y, p.ZZZ), S2 = mul29(p.Y, acc.ZZZ);           // <= 1.03
    Fp29<C> R = sub29<C, 2>(S2, S1); norm29(R);                                 // <= 3.03
    if (is_zero29(P)) {
        if (is_zero29(R)) dbl_xyzz29(acc); else inf = true;
        return;
    }
    const Fp29<C> PP = sqr29(P);                                                // <= 1.06
    const Fp29<C> PPP = mul29(P, PP), Q = mul29(U1, PP);                        // <= 1.02, 1.01
    Fp29<C> X3 = sub29<C, 2>(sub29<C, 2>(sub29<C, 2>(sqr29(R), PPP), Q), Q); norm29(X3);          // <= 1.06 + 6 = 7.06
    Fp29<C> T = sub29<C, 8>(Q, X3); norm29(T);                                  // <= 9.1
    const Fp29<C> Y3 = mul29_2(R, T, sub29<C, 2>(zero29<C>(), S1), PPP);        // (R T + (2p - S1) PPP) / R' + p <= (27.6 + 2.1) / 169 + 1 = 1.18
    acc.ZZ = mul29(mul29(acc.ZZ, p.ZZ), PP); acc.ZZZ = mul29(mul29(acc.ZZZ, p.ZZZ), PPP);
    acc.X = X3; acc.Y = Y3;
}


// one base-field element of a window table: canonical R-form -> canonical R'-form (x * 2^5 mod p), in place; all-zero stays all-zero
template <class C> __global__ void __launch_bounds__(256) k_table_to_r29(uint32_t* __restrict__ table, size_t n_elems) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_elems) return;
    Fp<C> v = fp_load<C>(table + i * C::N);
#pragma unroll
    for (int k = 0; k < 5; k++) v = fp_dbl(v);
    fp_store<C>(table + i * C::N, v);
}

// G1 accumulation over an R'-form window table (bases = table, infmask required)
template <class C, bool MERGE> __global__ void __launch_bounds__(256, 2)
k_msm_accum29(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ infmask, MsmShape sh, uint32_t skip, uint32_t cap, const uint32_t* __restrict__ counts,
              const uint32_t* __restrict__ starts, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ lane_g, const uint32_t* __restrict__ lane_sub,
              const uint32_t* __restrict__ meta, uint32_t* __restrict__ buckets, uint32_t* __restrict__ lane_partials, const uint32_t* __restrict__ prev_counts, int bucket_r29) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= meta[0]) return;
    const uint32_t g = lane_g[lane];
    const uint32_t cnt = counts[g];
    const uint32_t j = msm_key_lanes_log(msm_key(cnt, cap), cap);
    uint32_t lo = 0, hi = cnt;
    if (j) {
        const uint32_t chunk = (cnt + (1u << j) - 1) >> j;
        lo = min(cnt, lane_sub[lane] * chunk);
        hi = min(cnt, lo + chunk);
    }
    const uint32_t* list = sorted + starts[g];
    XYZZ29<C> acc;
    bool inf = true;
    if (MERGE) {
        if (prev_counts[g] && (!j || lane_sub[lane] == 0)) {
            const uint32_t* b = buckets + (size_t)g * 32;
            const uint4 z0 = reinterpret_cast<const uint4*>(b + 16)[0], z1 = reinterpret_cast<const uint4*>(b + 16)[1];
            if (z0.x | z0.y | z0.z | z0.w | z1.x | z1.y | z1.z | z1.w) {
                if (bucket_r29) {                                    // buckets kept in R'-form: shifts alone
                    acc.X = load29_packed<C>(b); acc.Y = load29_packed<C>(b + 8); acc.ZZ = load29_packed<C>(b + 16); acc.ZZZ = load29_packed<C>(b + 24);
                } else { acc.X = from_r256<C>(b); acc.Y = from_r256<C>(b + 8); acc.ZZ = from_r256<C>(b + 16); acc.ZZZ = from_r256<C>(b + 24); }
                inf = false;
            }
        }
    }
    uint32_t k = lo;
    // The gathered point stays in its packed form (4 x 16 bytes) until the iteration that consumes it: unpacking inside fetch would
    // wait for the loads at once and expose the gather latency that the software pipeline is there to hide.
    struct Raw { uint4 v[4]; };
    auto fetch = [&](uint32_t& e_out, Raw& r_out) -> bool {
        while (k < hi) {
            const uint32_t e = list[k++];
            uint32_t idx = e & 0x7fffffffu;
            if (idx < skip) continue;
            idx -= skip;
            if ((infmask[idx >> 5] >> (idx & 31)) & 1u) continue;
            const uint4* p = reinterpret_cast<const uint4*>(bases + (size_t)idx * 16);
            r_out.v[0] = p[0]; r_out.v[1] = p[1]; r_out.v[2] = p[2]; r_out.v[3] = p[3];
            e_out = e;
            return true;
        }
        return false;
    };
    auto unpack = [](const uint4& a, const uint4& b) { const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}; return unpack29<C>(w); };
    uint32_t e_next = 0;
    Raw r_next;
    bool have = fetch(e_next, r_next);
    while (have) {
        const uint32_t e = e_next;
        const Raw r = r_next;
        have = fetch(e_next, r_next);                       // the next gather is in flight during this addition
        Aff29<C> q;
        q.x = unpack(r.v[0], r.v[1]); q.y = unpack(r.v[2], r.v[3]);
        if (e >> 31) { q.y = sub29<C, 2>(zero29<C>(), q.y); norm29(q.y); }      // 2p - y
        madd29(acc, inf, q);
    }
    // lane partials of multi-lane buckets go to k_msm_tree in the reference's R-form; finished buckets stay in R'-form for the 29-bit row /
    // column sums (k_msm_rowcol_wave29) and for a later merge into the same buckets
    if (j) store_xyzz29<C, false>(lane_partials + (size_t)lane * 32, acc, inf);
    else if (bucket_r29) store_xyzz29<C, true>(buckets + (size_t)g * 32, acc, inf);
    else store_xyzz29<C, false>(buckets + (size_t)g * 32, acc, inf);
}

// Row / column sums of the 2-D bucket reduction (msm.cuh: k_msm_rowcol_wave) over R'-form buckets on 29-bit limbs: one wave per sum, 64 lanes add
// strided shares, then a 6-level tree through LDS in the same launch; the sums leave in the reference's R-form (k_msm_bitsums reads them).
template <class C> __global__ void __launch_bounds__(256)
k_msm_rowcol_wave29(MsmReduceBatch rb, uint32_t W, uint32_t nb, uint32_t rbits, uint32_t cbits, uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];                  // 256 lanes x 36 words
    __shared__ uint32_t inf_s[256];
    const uint32_t Cn = 1u << cbits, R = 1u << rbits;
    const uint32_t t = threadIdx.x, sub = t & 63u;
    const size_t n_out = (size_t)rb.njobs * W * 2 * Cn;
    for (size_t blk = blockIdx.x; blk * 4 < n_out; blk += gridDim.x) {
        const size_t gw = blk * 4 + (t >> 6);                    // sum index: ((job*W + w)*2 + kind)*C + i
        const bool valid = gw < n_out;
        const uint32_t i = (uint32_t)(gw & (Cn - 1)), kind = (uint32_t)(gw >> cbits) & 1u;
        const size_t jw = gw >> (cbits + 1);
        const uint32_t w = (uint32_t)(jw % W), job = (uint32_t)(jw / W);
        XYZZ29<C> acc;
        bool inf = true;
        if (valid) {
            const uint32_t* bk = rb.buckets[job];
            const uint32_t* cn = rb.counts[job];
            const uint32_t cnt = kind ? R : Cn;
            if (!kind && i >= R) { /* row index beyond the row count: empty sum */ }
            else for (uint32_t e = sub; e < cnt; e += 64) {
                const size_t g = (size_t)w * nb + (kind ? ((size_t)e << cbits) + i : ((size_t)i << cbits) + e);
                XYZZ29<C> p;
                if (cn[g] && load_xyzz29(p, bk + g * 32)) padd29(acc, inf, p);
            }
        }
        uint32_t* mine = lds + t;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            inf_s[t] = inf ? 1u : 0u;
            if (!inf) {
#pragma unroll
                for (int k = 0; k < 9; k++) { mine[k * 256] = acc.X.l[k]; mine[(9 + k) * 256] = acc.Y.l[k]; mine[(18 + k) * 256] = acc.ZZ.l[k]; mine[(27 + k) * 256] = acc.ZZZ.l[k]; }
            }
            __syncthreads();
            if ((t & (2 * d - 1)) == 0 && !inf_s[t + d]) {
                XYZZ29<C> o;
                const uint32_t* pn = lds + t + d;
#pragma unroll
                for (int k = 0; k < 9; k++) { o.X.l[k] = pn[k * 256]; o.Y.l[k] = pn[(9 + k) * 256]; o.ZZ.l[k] = pn[(18 + k) * 256]; o.ZZZ.l[k] = pn[(27 + k) * 256]; }
                padd29(acc, inf, o);
            }
            __syncthreads();
        }
        if (valid && sub == 0) store_xyzz29<C, false>(out + gw * 32, acc, inf);
    }
}

// ---- G2: Fq2 = Fq[u]/(u^2 + 1) over 29-bit limbs ------------------------------------------------------------------------------
// A product (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u is computed as TWO double products with one Montgomery reduction
// each (mul29_2; the minus sign is carried by a negated operand K p - b1): 2 x (162 + 81) MACs instead of Karatsuba's 3 x 171 plus its
// five additions / subtractions with their normalisations, and the outputs are plain products again (no offsets to track).
// A square is (a0 + a1)(a0 - a1) + 2 a0 a1 u.  All operands of mul29_2 must be normalised.
template <class C> struct F2x { Fp29<C> c0, c1; };
template <class C, int K> ZK_DEV Fp29<C> neg29(const Fp29<C>& a) { Fp29<C> r = sub29<C, K>(zero29<C>(), a); norm29(r); return r; }     // K p - a
// a * b, nb1 = K p - b.c1 supplied by the caller (often shared by several products)
template <class C> ZK_DEV F2x<C> f2mul(const F2x<C>& a, const F2x<C>& b, const Fp29<C>& nb1) {
    return F2x<C>{mul29_2(a.c0, b.c0, a.c1, nb1), mul29_2(a.c0, b.c1, a.c1, b.c0)};
}
// a^2 for components <= KB (the offset of the difference)
template <class C, int KB> ZK_DEV F2x<C> f2sqr(const F2x<C>& a) {
    Fp29<C> s = add29(a.c0, a.c1), d = sub29<C, KB>(a.c0, a.c1);
    norm29(d);                                                     // s: limbs < 2^30, d normalised
    Fp29<C> t = mul29(a.c0, a.c1);
    Fp29<C> c1 = add29(t, t);
    norm29(c1);
    return F2x<C>{mul29(s, d), c1};
}
template <class C, int K> ZK_DEV F2x<C> f2sub(const F2x<C>& a, const F2x<C>& b) { return F2x<C>{sub29<C, K>(a.c0, b.c0), sub29<C, K>(a.c1, b.c1)}; }   // not normalised
template <class C> ZK_DEV void f2norm(F2x<C>& a) { norm29(a.c0); norm29(a.c1); }
template <class C> ZK_DEV bool f2zero(const F2x<C>& a) { return is_zero29(a.c0) && is_zero29(a.c1); }

// XYZZ accumulator of a lane parked in LDS: word i of coordinate `coord` of lane t at ((coord * 18 + i) * T + t) (conflict-free)
template <class C, int T> struct LdsAcc29 {
    uint32_t* base;                                                // &lds[threadIdx.x]
    ZK_DEV void get(int coord, F2x<C>& v) const {
#pragma unroll
        for (int i = 0; i < 9; i++) { v.c0.l[i] = base[(coord * 18 + i) * T]; v.c1.l[i] = base[(coord * 18 + 9 + i) * T]; }
    }
    ZK_DEV void put(int coord, const F2x<C>& v) const {
#pragma unroll
        for (int i = 0; i < 9; i++) { base[(coord * 18 + i) * T] = v.c0.l[i]; base[(coord * 18 + 9 + i) * T] = v.c1.l[i]; }
    }
};
// Scheduling fence between Fq2-level operations of madd29_lds: the machine scheduler otherwise interleaves independent Fq2 products up to
// the register budget of the launch bounds (256 VGPRs + 25 spilled registers whose reloads wait on scratch); fenced, the kernel needs 208
// VGPRs and no scratch. Inside one Fq2 product the two component chains still overlap. (Same speed on MI355X, r02 A/B: the kernel is bound
// by integer issue, not by occupancy or spills — a 120-VGPR G1 variant at 4 waves per SIMD measured the same as well.)
#define ZK_SFENCE() __builtin_amdgcn_sched_barrier(0)
// acc += q over Fq2. Invariants (units of p, per component): X <= 8.4, Y <= 3.8, ZZ, ZZZ <= 1.1, all normalised. q.x canonical, q.y <= 2.
template <class C, int T> ZK_DEV void madd29_lds(const LdsAcc29<C, T>& A, bool& inf, const F2x<C>& qx, const F2x<C>& qy) {
    F2x<C> t;
    if (inf) {
        A.put(0, qx); A.put(1, qy);
        t.c0 = one29<C>(); t.c1 = zero29<C>();
        A.put(2, t); A.put(3, t);
        inf = false;
        return;
    }
    A.get(2, t);
    const F2x<C> U2 = f2mul(t, qx, neg29<C, 2>(qx.c1));                                // <= 1.1
    ZK_SFENCE();
    A.get(0, t);
    F2x<C> P = f2sub<C, 9>(U2, t); f2norm(P);                                           // X <= 8.4 < 9;  P <= 10.1
    A.get(3, t);
    const F2x<C> S2 = f2mul(t, qy, neg29<C, 3>(qy.c1));                                // <= 1.1
    ZK_SFENCE();
    A.get(1, t);
    F2x<C> R = f2sub<C, 4>(S2, t); f2norm(R);                                           // Y <= 3.8 < 4;  R <= 5.1
    if (f2zero(P)) {
        if (f2zero(R)) {
            // acc = 2 q (mdbl-2008-s-1, a = 0); rare: equal points in one bucket
            F2x<C> U{add29(qy.c0, qy.c0), add29(qy.c1, qy.c1)}; f2norm(U);             // <= 4
            const F2x<C> V = f2sqr<C, 5>(U);                                            // <= 2.2
            const Fp29<C> nV1 = neg29<C, 3>(V.c1);
            const F2x<C> W = f2mul(U, V, nV1), S = f2mul(qx, V, nV1);                   // <= 1.2
            const F2x<C> xx = f2sqr<C, 2>(qx);                                          // <= 2.1
            F2x<C> M{add29(add29(xx.c0, xx.c0), xx.c0), add29(add29(xx.c1, xx.c1), xx.c1)}; f2norm(M);      // <= 6.1
            F2x<C> X3 = f2sub<C, 2>(f2sub<C, 2>(f2sqr<C, 7>(M), S), S); f2norm(X3);    // <= 2.5 + 4 = 6.5
            F2x<C> Tt = f2sub<C, 7>(S, X3); f2norm(Tt);                                 // <= 8.2
            const F2x<C> MT = f2mul(M, Tt, neg29<C, 9>(Tt.c1)), Wy = f2mul(W, qy, neg29<C, 3>(qy.c1));       // <= 1.7, 1.1
            F2x<C> Y3 = f2sub<C, 2>(MT, Wy); f2norm(Y3);                                // <= 3.7
            A.put(0, X3); A.put(1, Y3); A.put(2, V); A.put(3, W);
        } else inf = true;
        return;
    }
    const F2x<C> PP = f2sqr<C, 11>(P);                                                  // <= 3.5
    ZK_SFENCE();
    const Fp29<C> nPP1 = neg29<C, 4>(PP.c1);
    const F2x<C> PPP = f2mul(P, PP, nPP1);                                              // <= 1.5
    ZK_SFENCE();
    const Fp29<C> nPPP1 = neg29<C, 2>(PPP.c1);
    A.get(2, t); A.put(2, f2mul(t, PP, nPP1));                                          // ZZ3 = ZZ1 * PP  <= 1.1
    ZK_SFENCE();
    A.get(3, t); A.put(3, f2mul(t, PPP, nPPP1));                                        // ZZZ3 = ZZZ1 * PPP
    ZK_SFENCE();
    A.get(0, t);
    const F2x<C> Q = f2mul(t, PP, nPP1);                                                // <= 1.4
    ZK_SFENCE();
    F2x<C> X3 = f2sub<C, 2>(f2sub<C, 2>(f2sub<C, 2>(f2sqr<C, 6>(R), PPP), Q), Q); f2norm(X3);      // <= 2.4 + 6 = 8.4
    A.put(0, X3);
    ZK_SFENCE();
    F2x<C> Tq = f2sub<C, 9>(Q, X3); f2norm(Tq);                                         // <= 10.4
    A.get(1, t);
    // Y3 = Tq R - Y1 PPP with ONE reduction per component (mul29_4, every operand normalised):
    //   c0 = Tq0 R0 + Tq1 (6p - R1) + Y0 (2p - PPP0) + Y1 PPP1      <= (53 + 62.4 + 7.6 + 5.7) / 169 + 1 = 1.8
    //   c1 = Tq0 R1 + Tq1 R0 + Y0 (2p - PPP1) + Y1 (2p - PPP0)      <= (53 + 53 + 7.6 + 7.6) / 169 + 1 = 1.8
    const Fp29<C> nR1 = neg29<C, 6>(R.c1), nPPP0 = neg29<C, 2>(PPP.c0);
    F2x<C> Y3;
    Y3.c0 = mul29_4(Tq.c0, R.c0, Tq.c1, nR1, t.c0, nPPP0, t.c1, PPP.c1);
    Y3.c1 = mul29_4(Tq.c0, R.c1, Tq.c1, R.c0, t.c0, nPPP1, t.c1, nPPP0);
    A.put(1, Y3);
}
template <class C, int T> ZK_DEV void store_xyzz29_lds(uint32_t* dst, const LdsAcc29<C, T>& A, bool inf) {
    if (inf) {
#pragma unroll
        for (int i = 0; i < 16; i++) reinterpret_cast<uint4*>(dst)[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    F2x<C> v;
#pragma unroll 1
    for (int cdn = 0; cdn < 4; cdn++) { A.get(cdn, v); store_r256(dst + cdn * 16, v.c0); store_r256(dst + cdn * 16 + 8, v.c1); }
}
// G2 accumulation over an R'-form window table: 256 lanes per block, accumulators in LDS (288 bytes per lane: two blocks per CU)
template <class C> __global__ void __launch_bounds__(256, 2)
k_msm_accum29_g2(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ infmask, MsmShape sh, uint32_t skip, uint32_t cap, const uint32_t* __restrict__ counts,
                 const uint32_t* __restrict__ starts, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ lane_g, const uint32_t* __restrict__ lane_sub,
                 const uint32_t* __restrict__ meta, uint32_t* __restrict__ buckets, uint32_t* __restrict__ lane_partials) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_acc29[];
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= meta[0]) return;
    const uint32_t g = lane_g[lane];
    const uint32_t cnt = counts[g];
    const uint32_t j = msm_key_lanes_log(msm_key(cnt, cap), cap);
    uint32_t lo = 0, hi = cnt;
    if (j) {
        const uint32_t chunk = (cnt + (1u << j) - 1) >> j;
        lo = min(cnt, lane_sub[lane] * chunk);
        hi = min(cnt, lo + chunk);
    }
    const uint32_t* list = sorted + starts[g];
    const LdsAcc29<C, 256> A{lds_acc29 + threadIdx.x};
    bool inf = true;
    auto unpack = [](const uint4& a, const uint4& b) { const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}; return unpack29<C>(w); };
    // Software pipeline as in the G1 kernel: the next point (packed, 8 x 16 bytes) is gathered while the current addition runs. On a box
    // with fast address translation this measures the same as the plain loop (the kernel is issue-bound); on boxes where random 128-byte
    // gathers over the 1.7 GB table are slow (r02: the same binary took 5.8 ms instead of 3.3 ms without it) two waves per SIMD cannot hide
    // the gather latency by themselves.
    uint32_t k = lo;
    struct Raw { uint4 v[8]; };
    auto fetch = [&](uint32_t& e_out, Raw& r_out) -> bool {
        while (k < hi) {
            const uint32_t e = list[k++];
            uint32_t idx = e & 0x7fffffffu;
            if (idx < skip) continue;
            idx -= skip;
            if ((infmask[idx >> 5] >> (idx & 31)) & 1u) continue;
            const uint4* p = reinterpret_cast<const uint4*>(bases + (size_t)idx * 32);
#pragma unroll
            for (int i = 0; i < 8; i++) r_out.v[i] = p[i];
            e_out = e;
            return true;
        }
        return false;
    };
    uint32_t e_next = 0;
    Raw r_next;
    bool have = fetch(e_next, r_next);
    while (have) {
        const uint32_t e = e_next;
        const Raw r = r_next;
        have = fetch(e_next, r_next);                           // the next gather is in flight during this addition
        F2x<C> qx{unpack(r.v[0], r.v[1]), unpack(r.v[2], r.v[3])}, qy{unpack(r.v[4], r.v[5]), unpack(r.v[6], r.v[7])};
        if (e >> 31) { qy.c0 = neg29<C, 2>(qy.c0); qy.c1 = neg29<C, 2>(qy.c1); }      // 2p - y
        madd29_lds<C, 256>(A, inf, qx, qy);
    }
    store_xyzz29_lds<C, 256>(j ? lane_partials + (size_t)lane * 64 : buckets + (size_t)g * 64, A, inf);
}

}  // namespace zkmi
