// snarkjs_amd/csrc/msm29.cuh — bucket accumulation over resident window tables on unsaturated limbs (field29.cuh): 9 x 29 bits for BN254,
// 14 x 28 bits for BLS12-381.
//
// Same job as k_msm_accum (msm.cuh: one lane per bucket or bucket share, XYZZ mixed additions over the sorted digit lists), but the
// field arithmetic of the hot loop runs on unsaturated limbs with lazy additions: BN254 10 products of ~207 instructions instead of ~290
// per mixed addition, BLS12-381 ~450 instead of ~620. Boundaries keep the library's formats: window tables are canonical N-word values in
// R'-form (k_table_to_r29 converts a table once, when it is built), lane partials are written in the reference's R-form (store_r256), so
// the lane-partial trees are unchanged; finished G1 buckets stay in R'-form for the row / column sums on the same limbs.
//
// Value bounds (units of p): a product of a and b comes back below a*b/(R'/p) + 1 with R'/p = 169 (BN254) or 2520 (BLS12-381). The bounds
// noted on each line are those of BN254, the tighter case; offsets K of the lazy subtractions are chosen from them. The point functions are
// __host__ __device__ like field29.cuh (tools/field29_hosttest.hip runs them on the CPU against Python big integers).
#pragma once
#include "field29.cuh"
#include "msm.cuh"

namespace zkmi {

#if defined(ZK29_SHADOW)
}  // namespace zkmi
#include <map>
namespace zkmi {
namespace b29 {
struct Parked { double bv, bl, bt; };
inline std::map<const void*, Parked>& parked() { static std::map<const void*, Parked> m; return m; }
}  // namespace b29
#endif
template <class C> struct Aff29 { Fp29<C> x, y; };
template <class C> struct XYZZ29 { Fp29<C> X, Y, ZZ, ZZZ; };          // invariants: X <= 7.3, Y <= 3.3, ZZ, ZZZ <= 1.1; all normalised

// doubling of an affine point (mdbl-2008-s-1, a = 0): the rare equal-points branch of madd29
template <class C> ZK_HD void dbl_affine29(XYZZ29<C>& r, const Aff29<C>& q) {
    Fp29<C> U = add29(q.y, q.y); norm29(U);                                     // <= 4
    const Fp29<C> V = mul29(U, U), W = mul29(U, V), S = mul29(q.x, V), xx = mul29(q.x, q.x);       // <= 1.1
    Fp29<C> M = add29(add29(xx, xx), xx); norm29(M);                            // <= 3.3
    Fp29<C> X3 = sub29<C, 2>(sub29<C, 2>(mul29(M, M), S), S); norm29(X3);       // <= 5.1
    const Fp29<C> T = sub29<C, 6>(S, X3);                                       // <= 7.1, limbs < 2^(B+2)
    Fp29<C> Y3 = sub29<C, 2>(mul29(M, T), mul29(W, q.y)); norm29(Y3);           // <= 3.2
    r.X = X3; r.Y = Y3; r.ZZ = V; r.ZZZ = W;
}
// acc += q (q affine, not the point at infinity, x canonical, y <= 2 normalised); inf = accumulator is the point at infinity.
// 8 products + 2 squarings (sqr29) with 9 reductions: the last two products share one (mul29_2).
template <class C> ZK_HD void madd29(XYZZ29<C>& acc, bool& inf, const Aff29<C>& q) {
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
    const Fp29<C> Y3 = mul29_2(R, T, sub29<C, 4>(zero29<C>(), acc.Y), PPP);     // 4p - Y1: limbs < 2^(B+1) (not normalised), as in f2mul
    acc.ZZ = mul29(acc.ZZ, PP); acc.ZZZ = mul29(acc.ZZZ, PPP);
    acc.X = X3; acc.Y = Y3;
}
// KEEP29: canonical words in R'-form (buckets that the row/column sums on the same limbs read back with shifts alone); else the reference's R-form
template <class C, bool KEEP29 = false> ZK_HD void store_xyzz29(uint32_t* dst, const XYZZ29<C>& a, bool inf) {
    constexpr int N = C::N;
    if (inf) {
#pragma unroll
        for (int i = 0; i < N; i++) reinterpret_cast<uint4*>(dst)[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    store_r256<C, KEEP29>(dst, a.X); store_r256<C, KEEP29>(dst + N, a.Y); store_r256<C, KEEP29>(dst + 2 * N, a.ZZ); store_r256<C, KEEP29>(dst + 3 * N, a.ZZZ);
}
// a point stored by store_xyzz29<C, true> (all-zero = infinity)
template <class C> ZK_HD bool load_xyzz29(XYZZ29<C>& a, const uint32_t* src) {
    constexpr int N = C::N;
    const uint4* q = reinterpret_cast<const uint4*>(src + 2 * N);
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N / 4; i++) { const uint4 z = q[i]; nz |= z.x | z.y | z.z | z.w; }
    if (!nz) return false;
    a.X = load29_packed<C>(src); a.Y = load29_packed<C>(src + N); a.ZZ = load29_packed<C>(src + 2 * N); a.ZZZ = load29_packed<C>(src + 3 * N);
    return true;
}
// acc = 2 acc for a general XYZZ accumulator (dbl-2008-s-1, a = 0): the rare equal-points branch of padd29. Same invariants as madd29.
template <class C> ZK_HD void dbl_xyzz29(XYZZ29<C>& r) {
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
// The coordinates of p are fetched on demand through ld(k), k = 0..3 = X, Y, ZZ, ZZZ (ZZ and ZZZ twice): p never sits in registers as a
// whole — with 14-limb coordinates the two points, the intermediate values and a product's scratch do not fit 256 VGPRs together (the first
// BLS12-381 version took 306 registers, one wave per SIMD).
template <class C, class Ld> ZK_HD void padd29_ld(XYZZ29<C>& acc, bool& inf, Ld ld) {
    if (inf) { acc.X = ld(0); acc.Y = ld(1); acc.ZZ = ld(2); acc.ZZZ = ld(3); inf = false; return; }
    const Fp29<C> U1 = mul29(acc.X, ld(2)), U2 = mul29(ld(0), acc.ZZ);          // <= 1.05
    Fp29<C> P = sub29<C, 2>(U2, U1); norm29(P);                                 // <= 3.05
    const Fp29<C> S1 = mul29(acc.Y, ld(3)), S2 = mul29(ld(1), acc.ZZZ);         // <= 1.03
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
    acc.ZZ = mul29(mul29(acc.ZZ, ld(2)), PP); acc.ZZZ = mul29(mul29(acc.ZZZ, ld(3)), PPP);
    acc.X = X3; acc.Y = Y3;
}
template <class C> ZK_HD void padd29(XYZZ29<C>& acc, bool& inf, const XYZZ29<C>& p) {
    padd29_ld(acc, inf, [&](int k) -> Fp29<C> { return k == 0 ? p.X : (k == 1 ? p.Y : (k == 2 ? p.ZZ : p.ZZZ)); });
}
// all-zero ZZ words = the point at infinity (store_xyzz29<C, true>)
template <class C> ZK_HD bool xyzz29_words_inf(const uint32_t* src) {
    const uint4* q = reinterpret_cast<const uint4*>(src + 2 * C::N);
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < C::N / 4; i++) { const uint4 z = q[i]; nz |= z.x | z.y | z.z | z.w; }
    return nz == 0;
}


// the same for an Fq2 point (ZZ = words 4N .. 6N of 8N)
template <class C> ZK_HD bool xyzz29_words_inf_g2(const uint32_t* src) {
    const uint4* q = reinterpret_cast<const uint4*>(src + 4 * C::N);
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < C::N / 2; i++) { const uint4 z = q[i]; nz |= z.x | z.y | z.z | z.w; }
    return nz == 0;
}

// one base-field element of a window table: canonical R-form -> canonical R'-form (x * 2^5 mod p, 2^8 for BLS12-381), in place; all-zero
// stays all-zero
template <class C> __global__ void __launch_bounds__(256) k_table_to_r29(uint32_t* __restrict__ table, size_t n_elems) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_elems) return;
    Fp<C> v = fp_load<C>(table + i * C::N);
#pragma unroll
    for (int k = 0; k < r29_shift<C>(); k++) v = fp_dbl(v);
    fp_store<C>(table + i * C::N, v);
}

// Sorted digit lists are read FOUR entries (16 bytes) at a time: a lane comes back for its next entry only after a whole mixed addition
// (2 200 .. 13 000 instructions), by which time the streaming gathers have evicted the line — a 4-byte read per entry cost a whole 64-byte
// sector each (r02: 54 MB of lists became 0.87 GB of HBM reads per 2^20 MSM, half of the accumulation's traffic). The group of four is
// aligned in the flat `sorted` array, so the first and last group of a list may hold neighbours' entries, which are never selected.
struct ListReader {
    const uint4* sorted4;
    uint32_t a, a_end, at = 0xffffffffu;
    uint4 buf;
    ZK_DEV ListReader(const uint32_t* sorted, uint32_t first, uint32_t last) : sorted4(reinterpret_cast<const uint4*>(sorted)), a(first), a_end(last), buf(make_uint4(0, 0, 0, 0)) {}
    ZK_DEV bool more() const { return a < a_end; }
    ZK_DEV uint32_t next() {
        const uint32_t grp = a >> 2, sel = a & 3u;
        if (grp != at) { buf = sorted4[grp]; at = grp; }
        a++;
        return sel == 0 ? buf.x : (sel == 1 ? buf.y : (sel == 2 ? buf.z : buf.w));
    }
};
// r06: one list entry taken SPECULATIVELY. The accumulation loops used to read a list entry, wait for the word of the infinity bitmap it points at, and only
// then issue the gather of the table entry — a dependent wait (an L2 round trip) in front of every mixed addition, in the wave that is about to issue it.
// Now the bitmap word and the table entry are requested together and the bit is looked at one addition later, when the point is consumed: by then both
// have arrived. Entries of points at infinity (and entries below `skip`, which belong to another slice of the table) are rare in a list — the sort drops
// the bases a key knows to be zero — and cost one pass of the catch-up loop, which waits the way the old code always did.
// Measured (profiles/r06_spec_gather_ab.txt, one box, product against a -DZK_SPEC_GATHER=0 build, interleaved, two runs each): G1 accumulation 1.072 / 1.074 against
// 1.079 / 1.099 ms, 14-limb G2 6.96 / 7.05 against 7.09 / 7.25 ms, but BN254 G2 2.90 / 2.96 against 2.86 / 2.85 ms (three more live registers at 245 of 256) —
// so the G1 kernels and the 14-limb G2 kernel take it, BN254 G2 keeps the waiting loop. End to end nothing moves by more than the run-to-run spread: the
// kernels are bound by integer issue, not by this wait.
#ifndef ZK_SPEC_GATHER
#define ZK_SPEC_GATHER 1
#endif
#ifndef ZK_SPEC_GATHER_G2
#define ZK_SPEC_GATHER_G2(C) (ZK_SPEC_GATHER && Lim29<C>::NL > 9)
#endif
struct SpecEntry { uint32_t e, m, s; };                            // list entry, bitmap word, bit number in it (32: not in this slice of the table)
ZK_DEV bool spec_dead(const SpecEntry& x) { return x.s >= 32u || ((x.m >> x.s) & 1u); }
// an affine coordinate (C::N words = C::N / 4 vectors) of a gathered table entry -> limbs
template <class C> ZK_DEV Fp29<C> unpack29_v(const uint4* v) {
    uint32_t w[C::N];
#pragma unroll
    for (int i = 0; i < C::N / 4; i++) { w[4 * i] = v[i].x; w[4 * i + 1] = v[i].y; w[4 * i + 2] = v[i].z; w[4 * i + 3] = v[i].w; }
    return unpack29<C>(w);
}

// G1 accumulation over an R'-form window table (bases = table, infmask required)
template <class C, bool MERGE> __global__ void __launch_bounds__(256, 2)
k_msm_accum29(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ infmask, MsmShape sh, uint32_t skip, uint32_t cap, const uint32_t* __restrict__ counts,
              const uint32_t* __restrict__ starts, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ lane_g, const uint32_t* __restrict__ lane_sub,
              const uint32_t* __restrict__ meta, uint32_t* __restrict__ buckets, uint32_t* __restrict__ lane_partials, const uint32_t* __restrict__ prev_counts, int bucket_r29) {
    constexpr int N = C::N, PW = 4 * N, AV = N / 2;                 // words per XYZZ point; 16-byte vectors per affine table entry
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
    XYZZ29<C> acc;
    bool inf = true;
    if (MERGE) {
        if (prev_counts[g] && (!j || lane_sub[lane] == 0)) {
            const uint32_t* b = buckets + (size_t)g * PW;
            if (bucket_r29) inf = !load_xyzz29(acc, b);              // buckets kept in R'-form: shifts alone
            else {
                uint32_t nz = 0;
#pragma unroll
                for (int i = 0; i < N / 4; i++) { const uint4 z = reinterpret_cast<const uint4*>(b + 2 * N)[i]; nz |= z.x | z.y | z.z | z.w; }
                if (nz) { acc.X = from_r256<C>(b); acc.Y = from_r256<C>(b + N); acc.ZZ = from_r256<C>(b + 2 * N); acc.ZZZ = from_r256<C>(b + 3 * N); inf = false; }
            }
        }
    }
    const uint32_t s0 = starts[g];
    ListReader list(sorted, s0 + lo, s0 + hi);
    // The gathered point stays in its packed form (AV x 16 bytes) until the iteration that consumes it: unpacking inside fetch would
    // wait for the loads at once and expose the gather latency that the software pipeline is there to hide.
    struct Raw { uint4 v[AV]; };
    auto fetch = [&](uint32_t& e_out, Raw& r_out) -> bool {
        while (list.more()) {
            const uint32_t e = list.next();
            uint32_t idx = e & 0x7fffffffu;
            if (idx < skip) continue;
            idx -= skip;
            if ((infmask[idx >> 5] >> (idx & 31)) & 1u) continue;
            const uint4* p = reinterpret_cast<const uint4*>(bases + (size_t)idx * (2 * N));
#pragma unroll
            for (int i = 0; i < AV; i++) r_out.v[i] = p[i];
            e_out = e;
            return true;
        }
        return false;
    };
#if ZK_SPEC_GATHER
    auto fetch_spec = [&](SpecEntry& x, Raw& r_out) -> bool {
        if (!list.more()) return false;
        const uint32_t e = list.next();
        uint32_t idx = e & 0x7fffffffu;
        const bool mine = idx >= skip;
        idx = mine ? idx - skip : 0u;
        x.e = e; x.s = mine ? (idx & 31u) : 32u;
        x.m = infmask[idx >> 5];
        const uint4* p = reinterpret_cast<const uint4*>(bases + (size_t)idx * (2 * N));
#pragma unroll
        for (int i = 0; i < AV; i++) r_out.v[i] = p[i];
        return true;
    };
    SpecEntry x_next{0, 0, 32};
    Raw r_next;
    bool have = fetch_spec(x_next, r_next);
    while (have) {
        SpecEntry x = x_next;
        Raw r = r_next;
        have = fetch_spec(x_next, r_next);                  // the next bitmap word and gather are in flight during this addition
        while (spec_dead(x) && have) { x = x_next; r = r_next; have = fetch_spec(x_next, r_next); }
        if (spec_dead(x)) break;
        Aff29<C> q;
        q.x = unpack29_v<C>(r.v); q.y = unpack29_v<C>(r.v + AV / 2);
        if (x.e >> 31) { q.y = sub29<C, 2>(zero29<C>(), q.y); norm29(q.y); }    // 2p - y
        madd29(acc, inf, q);
    }
#else
    uint32_t e_next = 0;
    Raw r_next;
    bool have = fetch(e_next, r_next);
    while (have) {
        const uint32_t e = e_next;
        const Raw r = r_next;
        have = fetch(e_next, r_next);                       // the next gather is in flight during this addition
        Aff29<C> q;
        q.x = unpack29_v<C>(r.v); q.y = unpack29_v<C>(r.v + AV / 2);
        if (e >> 31) { q.y = sub29<C, 2>(zero29<C>(), q.y); norm29(q.y); }      // 2p - y
        madd29(acc, inf, q);
    }
#endif
    // lane partials of multi-lane buckets go to k_msm_tree in the reference's R-form; finished buckets stay in R'-form for the row /
    // column sums on the same limbs (k_msm_rowcol_wave29) and for a later merge into the same buckets
    if (j) store_xyzz29<C, false>(lane_partials + (size_t)lane * PW, acc, inf);
    else if (bucket_r29) store_xyzz29<C, true>(buckets + (size_t)g * PW, acc, inf);
    else store_xyzz29<C, false>(buckets + (size_t)g * PW, acc, inf);
}

// Row / column sums of the 2-D bucket reduction (msm.cuh: k_msm_rowcol_wave) over R'-form buckets on unsaturated limbs: one wave per sum, 64
// lanes add strided shares, then a 6-level tree through LDS in the same launch; the sums leave in the reference's R-form (k_msm_bitsums reads them).
// waves per SIMD the register budget is held to: 3 with 9-limb coordinates (168 VGPRs), 2 with 14-limb ones (256)
template <class C> __global__ void __launch_bounds__(256, Lim29<C>::NL <= 9 ? 3 : 2)
k_msm_rowcol_wave29(MsmReduceBatch rb, uint32_t W, uint32_t nb, uint32_t rbits, uint32_t cbits, uint32_t* __restrict__ out) {
    constexpr int NL = Lim29<C>::NL, PW = 4 * C::N;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];                  // 256 lanes x 4 NL words
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
                const uint32_t* src = bk + g * PW;
                if (cn[g] && !xyzz29_words_inf<C>(src)) padd29_ld(acc, inf, [&](int k) { return load29_packed<C>(src + k * C::N); });
            }
        }
        uint32_t* mine = lds + t;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            inf_s[t] = inf ? 1u : 0u;
            if (!inf) {
#pragma unroll
                for (int k = 0; k < NL; k++) { mine[k * 256] = acc.X.l[k]; mine[(NL + k) * 256] = acc.Y.l[k]; mine[(2 * NL + k) * 256] = acc.ZZ.l[k]; mine[(3 * NL + k) * 256] = acc.ZZZ.l[k]; }
            }
            __syncthreads();
            if ((t & (2 * d - 1)) == 0 && !inf_s[t + d]) {
                const uint32_t* pn = lds + t + d;
                padd29_ld(acc, inf, [&](int k) { Fp29<C> v;
#pragma unroll
                    for (int i = 0; i < NL; i++) v.l[i] = pn[(k * NL + i) * 256];
                    return v; });
            }
            __syncthreads();
        }
        if (valid && sub == 0) store_xyzz29<C, false>(out + gw * PW, acc, inf);
    }
}

// ---- G2: Fq2 = Fq[u]/(u^2 + 1) over unsaturated limbs --------------------------------------------------------------------------
// A product (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u is computed as TWO double products with one Montgomery reduction
// each (mul29_2; the minus sign is carried by a negated operand K p - b1): BN254 2 x (162 + 81) MACs instead of Karatsuba's 3 x 171 plus its
// five additions / subtractions with their normalisations, and the outputs are plain products again (no offsets to track).
// A square is (a0 + a1)(a0 - a1) + 2 a0 a1 u.  All operands of mul29_2 must be normalised.
template <class C> struct F2x { Fp29<C> c0, c1; };
template <class C, int K> ZK_HD Fp29<C> neg29(const Fp29<C>& a) { Fp29<C> r = sub29<C, K>(zero29<C>(), a); norm29(r); return r; }     // K p - a
// Compact<C> (field29.cuh): the Fq2 product and square are CALLED — one copy of each in the code object instead of one per site (~15 per
// addition); operands and result travel in registers by the AMDGPU calling convention, the excess over 32 argument registers through the stack
#define ZK_F2_CALLS(C) (IsCompact<C>::value)
// The negated component is formed INSIDE the call with one generous offset (16 p - b.c1: every site's b.c1 is far below that, and with
// R' / p = 2^11 on the 14-limb curve the larger operand costs nothing in the bounds of the double product): 4 NL argument words instead of 5 NL
template <class C> __host__ __device__ ZK_NOINLINE_DEV F2x<C> f2mul_call(F2x<C> a, F2x<C> b) {
    static_assert(Lim29<C>::NL > 9, "called Fq2 product: 14-limb curve only (offset 16 p needs its headroom)");
    const Fp29<C> nb1 = neg29<C, 16>(b.c1);
    return F2x<C>{mul29_2_inl(a.c0, b.c0, a.c1, nb1), mul29_2_inl(a.c0, b.c1, a.c1, b.c0)};
}
// a * b, nb1 = K p - b.c1 supplied by the caller (often shared by several products)
template <class C> ZK_HD F2x<C> f2mul(const F2x<C>& a, const F2x<C>& b, const Fp29<C>& nb1) {
    if constexpr (ZK_F2_CALLS(C)) return f2mul_call<C>(a, b);
    else return F2x<C>{mul29_2(a.c0, b.c0, a.c1, nb1), mul29_2(a.c0, b.c1, a.c1, b.c0)};
}
// a^2 for components <= KB (the offset of the difference)
template <class C, int KB> ZK_HD F2x<C> f2sqr(const F2x<C>& a) {
    Fp29<C> s = add29(a.c0, a.c1), d = sub29<C, KB>(a.c0, a.c1);
    norm29(d);                                                     // s: limbs < 2^(B+1), d normalised
    // Compact<C>: two mul29 CALLS, each with its 2 NL argument words in registers (one call for the whole square would push half of its four
    // operands through the stack)
    Fp29<C> t = mul29(a.c0, a.c1);
    Fp29<C> c1 = add29(t, t);
    norm29(c1);
    return F2x<C>{mul29(s, d), c1};
}
template <class C, int K> ZK_HD F2x<C> f2sub(const F2x<C>& a, const F2x<C>& b) { return F2x<C>{sub29<C, K>(a.c0, b.c0), sub29<C, K>(a.c1, b.c1)}; }   // not normalised
template <class C> ZK_HD void f2norm(F2x<C>& a) { norm29(a.c0); norm29(a.c1); }
template <class C> ZK_HD bool f2zero(const F2x<C>& a) { return is_zero29(a.c0) && is_zero29(a.c1); }

// XYZZ accumulator of a lane parked in LDS: word i of component `comp` of coordinate `coord` of lane t at (((2 coord + comp) EW + i) T + t)
// (conflict-free). PACK = false: EW = NL limbs as they are (BN254: 288 bytes per lane, two 256-lane blocks per CU). PACK = true: EW = N packed
// words (BLS12-381: 384 instead of 448 bytes per lane — three 128-lane blocks per CU instead of two; every parked value is normalised and
// below 2^384 = 9.7 p, the shifts cost ~5 % of the addition's instructions and buy half a wave per SIMD).
template <class C, int T, bool PACK> struct LdsAcc29 {
    static constexpr int NL = Lim29<C>::NL, EW = PACK ? C::N : NL;
    uint32_t* base;                                                // &lds[threadIdx.x]
    ZK_HD void get1(int slot, Fp29<C>& v) const {
        if constexpr (PACK) {
            uint32_t w[C::N];
#pragma unroll
            for (int i = 0; i < C::N; i++) w[i] = base[(slot * EW + i) * T];
            v = unpack29<C>(w);
        } else {
#pragma unroll
            for (int i = 0; i < NL; i++) v.l[i] = base[(slot * EW + i) * T];
        }
#if defined(ZK29_SHADOW)
        {   // the bounds a parked value was stored with travel beside the "LDS" array of the host test (keyed by the slot's first word)
            auto it = b29::parked().find(&base[(size_t)slot * EW * T]);
            b29::need(it != b29::parked().end(), "LdsAcc29: slot read before it was written", slot, 0);
            if (it != b29::parked().end()) { v.bv = it->second.bv; v.bl = it->second.bl; v.bt = it->second.bt; }
        }
#endif
    }
    ZK_HD void put1(int slot, const Fp29<C>& v) const {
#if defined(ZK29_SHADOW)
        {
            b29::need(b29::normalised(v), "LdsAcc29: a parked value is not normalised", v.bl, b29::lowmax<C>());
            b29::Parked pk; pk.bv = v.bv; pk.bl = v.bl; pk.bt = v.bt;
            b29::parked()[&base[(size_t)slot * EW * T]] = pk;
        }
#endif
        if constexpr (PACK) {
            uint32_t w[C::N];
            pack29<C>(w, v);
#pragma unroll
            for (int i = 0; i < C::N; i++) base[(slot * EW + i) * T] = w[i];
        } else {
#pragma unroll
            for (int i = 0; i < NL; i++) base[(slot * EW + i) * T] = v.l[i];
        }
    }
    ZK_HD void get(int coord, F2x<C>& v) const { get1(2 * coord, v.c0); get1(2 * coord + 1, v.c1); }
    ZK_HD void put(int coord, const F2x<C>& v) const { put1(2 * coord, v.c0); put1(2 * coord + 1, v.c1); }
};
// Scheduling fence between Fq2-level operations of madd29_lds: the machine scheduler otherwise interleaves independent Fq2 products up to
// the register budget of the launch bounds (256 VGPRs + 25 spilled registers whose reloads wait on scratch); fenced, the kernel needs 208
// VGPRs and no scratch. Inside one Fq2 product the two component chains still overlap. (Same speed on MI355X, r02 A/B: the kernel is bound
// by integer issue, not by occupancy or spills — a 120-VGPR G1 variant at 4 waves per SIMD measured the same as well.)
#ifndef ZK_Y3_SPLIT
#define ZK_Y3_SPLIT(C) (Lim29<C>::NL > 9 || IsCompact<C>::value)
#endif
#ifndef ZK_G2_PREFETCH
#define ZK_G2_PREFETCH(C) true
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_SFENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ZK_SFENCE() ((void)0)
#endif
// acc += q over Fq2. Invariants (units of p, per component): X <= 8.4, Y <= 3.8, ZZ <= 2.2, ZZZ <= 1.2, all normalised. q.x canonical, q.y <= 2.
// (ZZ, ZZZ <= 1.1 on the main path; the doubling branch leaves ZZ = V — a square, whose c1 is a doubled product: up to 2.19 — and ZZZ = W as they
// come. Nothing here ever NEGATES a component of ZZ or ZZZ, so the wider bound costs nothing; the bucket leaves through store_r256, canonical. The
// reduction kernels' padd29_lds does negate them with offset 2 p: its operands are canonical words or results of padd29_lds / dbl29_lds, where ZZ and
// ZZZ are products, <= 1.1. Both contracts are checked for the worst case by tests/test_field29_host.py::test_worst_case_bounds_of_the_point_formulas,
// which found the difference.)
template <class C, class Acc> ZK_HD void madd29_lds(const Acc& A, bool& inf, const F2x<C>& qx, const F2x<C>& qy) {
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
    // Y3 = Tq R - Y1 PPP with ONE reduction per component (mul29_4, every operand normalised):
    //   c0 = Tq0 R0 + Tq1 (6p - R1) + Y0 (2p - PPP0) + Y1 PPP1      <= (53 + 62.4 + 7.6 + 5.7) / 169 + 1 = 1.8
    //   c1 = Tq0 R1 + Tq1 R0 + Y0 (2p - PPP1) + Y1 (2p - PPP0)      <= (53 + 53 + 7.6 + 7.6) / 169 + 1 = 1.8
    // Y3_SPLIT (14-limb moduli, where Tq, R, Y1, PPP and the three negations together exceed the register file next to the prefetched point):
    // the same sum as two 2-product sums per component — Tq R first, then Y1 and PPP — with two reductions: +392 MACs, half the live operands
    const Fp29<C> nR1 = neg29<C, 6>(R.c1);
    F2x<C> Y3;
    if constexpr (ZK_Y3_SPLIT(C)) {
        const F2x<C> TR{mul29_2(Tq.c0, R.c0, Tq.c1, nR1), mul29_2(Tq.c0, R.c1, Tq.c1, R.c0)};     // <= (53 + 62.4) / 169 + 1 = 1.7
        ZK_SFENCE();
        const Fp29<C> nPPP0 = neg29<C, 2>(PPP.c0);
        A.get(1, t);
        Y3.c0 = add29(TR.c0, mul29_2(t.c0, nPPP0, t.c1, PPP.c1));                       // <= 1.7 + 1.1
        Y3.c1 = add29(TR.c1, mul29_2(t.c0, nPPP1, t.c1, nPPP0));
        f2norm(Y3);
    } else {
        const Fp29<C> nPPP0 = neg29<C, 2>(PPP.c0);
        A.get(1, t);
        Y3.c0 = mul29_4(Tq.c0, R.c0, Tq.c1, nR1, t.c0, nPPP0, t.c1, PPP.c1);
        Y3.c1 = mul29_4(Tq.c0, R.c1, Tq.c1, R.c0, t.c0, nPPP1, t.c1, nPPP0);
    }
    A.put(1, Y3);
}
template <class C, class Acc, bool KEEP29 = false> ZK_HD void store_xyzz29_lds(uint32_t* dst, const Acc& A, bool inf) {
    constexpr int N = C::N;
    if (inf) {
#pragma unroll
        for (int i = 0; i < 2 * N; i++) reinterpret_cast<uint4*>(dst)[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    F2x<C> v;
#pragma unroll 1
    for (int cdn = 0; cdn < 4; cdn++) { A.get(cdn, v); store_r256<C, KEEP29>(dst + cdn * 2 * N, v.c0); store_r256<C, KEEP29>(dst + cdn * 2 * N + N, v.c1); }
}

// ---- the same accumulation with a JACOBIAN accumulator (X, Y, Z: x = X / Z^2, y = Y / Z^3) for 14-limb moduli --------------------------
// The XYZZ accumulator of an Fq2 point is 4 x 2 x 12 packed words = 384 bytes per lane: three 128-lane blocks per CU, 1.5 waves per SIMD.
// Three coordinates are 288 bytes: FOUR blocks, two waves per SIMD on every SIMD — where a dependent chain of 14-limb products runs at
// 73 G/s instead of the 59 G/s average of a 2 + 1 split (tools/fieldbench29, profiles/r03_fieldbench29.txt). The price is one squaring:
// 8M + 3S (Z^2 first) instead of 8M + 2S, +7 % multiply-accumulates. Coordinates in LDS slots 0 = X, 1 = Y, 2 = Z.
//   ZZ = Z1^2, ZZZ = Z1 ZZ, U2 = x2 ZZ, S2 = y2 ZZZ, H = U2 - X1, R = S2 - Y1, HH = H^2, HHH = H HH, V = X1 HH,
//   X3 = R^2 - HHH - 2 V, Y3 = R (V - X3) - Y1 HHH, Z3 = Z1 H
// Invariants (units of p, per component): X <= 8.4, Y <= 3.8, Z <= 1.1, all normalised. q.x canonical, q.y <= 2. Offsets as in madd29_lds.
// requery(qx, qy): fetches q again for the rare doubling branch — q is dead after S2 on the main path, which frees 56 registers at the
// point where the live set peaks (H, R, the prefetched next point, a product's scratch)
template <class C, class Acc, class Rq> ZK_HD void madd29_jac_lds(const Acc& A, bool& inf, const F2x<C>& qx_in, const F2x<C>& qy_in, Rq requery) {
    F2x<C> t;
    const F2x<C>&qx = qx_in, &qy = qy_in;
    if (inf) {
        A.put(0, qx); A.put(1, qy);
        t.c0 = one29<C>(); t.c1 = zero29<C>();
        A.put(2, t);
        inf = false;
        return;
    }
    A.get(2, t);
    const F2x<C> ZZ = f2sqr<C, 2>(t);                                                   // <= 2.1 (second component: 2 a0 a1)
    ZK_SFENCE();
    const Fp29<C> nZZ1 = neg29<C, 3>(ZZ.c1);
    const F2x<C> ZZZ = f2mul(t, ZZ, nZZ1);                                              // <= 1.1
    ZK_SFENCE();
    const F2x<C> U2 = f2mul(qx, ZZ, nZZ1);                                              // <= 1.1
    ZK_SFENCE();
    A.get(0, t);
    F2x<C> H = f2sub<C, 9>(U2, t); f2norm(H);                                           // X <= 8.4 < 9;  H <= 10.1
    const F2x<C> S2 = f2mul(qy, ZZZ, neg29<C, 2>(ZZZ.c1));                              // <= 1.1
    ZK_SFENCE();
    A.get(1, t);
    F2x<C> R = f2sub<C, 4>(S2, t); f2norm(R);                                           // Y <= 3.8 < 4;  R <= 5.1
    if (f2zero(H)) {
        if (f2zero(R)) {
            // acc = 2 q (mdbl-2008-s-1, a = 0): XYZZ (X3, Y3, U^2, U^3) is the Jacobian point (X3, Y3, Z = U), U = 2 y
            F2x<C> qx, qy;
            requery(qx, qy);
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
            // Z3 = U = 2 y is up to 4 p: one product by one brings it back under the invariant (Z <= 1.1) the next addition's offsets assume
            F2x<C> one; one.c0 = one29<C>(); one.c1 = zero29<C>();
            A.put(0, X3); A.put(1, Y3); A.put(2, f2mul(U, one, zero29<C>()));
        } else inf = true;
        return;
    }
    const F2x<C> HH = f2sqr<C, 11>(H);                                                  // <= 3.5
    ZK_SFENCE();
    const Fp29<C> nHH1 = neg29<C, 4>(HH.c1);
    const F2x<C> HHH = f2mul(H, HH, nHH1);                                              // <= 1.5
    ZK_SFENCE();
    A.get(2, t); A.put(2, f2mul(t, H, neg29<C, 11>(H.c1)));                             // Z3 = Z1 H  <= 1.1
    ZK_SFENCE();
    A.get(0, t);
    const F2x<C> V = f2mul(t, HH, nHH1);                                                // <= 1.4
    ZK_SFENCE();
    F2x<C> X3 = f2sub<C, 2>(f2sub<C, 2>(f2sub<C, 2>(f2sqr<C, 6>(R), HHH), V), V); f2norm(X3);     // <= 2.4 + 6 = 8.4
    A.put(0, X3);
    ZK_SFENCE();
    F2x<C> Tq = f2sub<C, 9>(V, X3); f2norm(Tq);                                         // <= 10.4
    // Y3 = Tq R - Y1 HHH as two 2-product sums per component (see madd29_lds: the split form, half the live operands)
    const Fp29<C> nR1 = neg29<C, 6>(R.c1);
    const F2x<C> TR{mul29_2(Tq.c0, R.c0, Tq.c1, nR1), mul29_2(Tq.c0, R.c1, Tq.c1, R.c0)};         // <= (53 + 62.4) / 169 + 1 = 1.7
    ZK_SFENCE();
    const Fp29<C> nHHH0 = neg29<C, 2>(HHH.c0), nHHH1 = neg29<C, 2>(HHH.c1);
    A.get(1, t);
    F2x<C> Y3;
    Y3.c0 = add29(TR.c0, mul29_2(t.c0, nHHH0, t.c1, HHH.c1));                           // <= 1.7 + 1.1
    Y3.c1 = add29(TR.c1, mul29_2(t.c0, nHHH1, t.c1, nHHH0));
    f2norm(Y3);
    A.put(1, Y3);
}
// Jacobian (X, Y, Z) parked in LDS -> the XYZZ words of the bucket arrays: ZZ = Z^2, ZZZ = Z ZZ (once per lane)
template <class C, class Acc, bool KEEP29 = false> ZK_HD void store_jac29_lds(uint32_t* dst, const Acc& A, bool inf) {
    constexpr int N = C::N;
    if (inf) {
#pragma unroll
        for (int i = 0; i < 2 * N; i++) reinterpret_cast<uint4*>(dst)[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    F2x<C> v;
    A.get(0, v); store_r256<C, KEEP29>(dst, v.c0); store_r256<C, KEEP29>(dst + N, v.c1);
    A.get(1, v); store_r256<C, KEEP29>(dst + 2 * N, v.c0); store_r256<C, KEEP29>(dst + 3 * N, v.c1);
    A.get(2, v);
    const F2x<C> ZZ = f2sqr<C, 2>(v);
    const F2x<C> ZZZ = f2mul(v, ZZ, neg29<C, 3>(ZZ.c1));
    store_r256<C, KEEP29>(dst + 4 * N, ZZ.c0); store_r256<C, KEEP29>(dst + 5 * N, ZZ.c1);
    store_r256<C, KEEP29>(dst + 6 * N, ZZZ.c0); store_r256<C, KEEP29>(dst + 7 * N, ZZZ.c1);
}
// G2 accumulation over an R'-form window table, accumulators in LDS: BN254 256 lanes per block, XYZZ accumulators as they are (288 bytes per
// lane: two blocks per CU); BLS12-381 128 lanes per block, packed JACOBIAN accumulators (288 bytes per lane: four blocks per CU)
template <class C> struct Accum29G2 {
    static constexpr int T = MsmAccumBlock<Fp2<C>>::value;
    static constexpr bool PACK = C::N > 8, JAC = PACK;
    typedef LdsAcc29<C, T, PACK> Acc;
    static constexpr size_t lds_bytes = (size_t)T * (JAC ? 6 : 8) * Acc::EW * 4;
    template <class A, class Rq> ZK_HD static void madd(const A& acc, bool& inf, const F2x<C>& qx, const F2x<C>& qy, Rq requery) {
        if constexpr (JAC) madd29_jac_lds<C>(acc, inf, qx, qy, requery); else madd29_lds<C>(acc, inf, qx, qy);
    }
    template <bool KEEP29, class A> ZK_HD static void store(uint32_t* dst, const A& acc, bool inf) {
        if constexpr (JAC) store_jac29_lds<C, A, KEEP29>(dst, acc, inf); else store_xyzz29_lds<C, A, KEEP29>(dst, acc, inf);
    }
};
template <class C> __global__ void __launch_bounds__(Accum29G2<C>::T, 2)
k_msm_accum29_g2(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ infmask, MsmShape sh, uint32_t skip, uint32_t cap, const uint32_t* __restrict__ counts,
                 const uint32_t* __restrict__ starts, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ lane_g, const uint32_t* __restrict__ lane_sub,
                 const uint32_t* __restrict__ meta, uint32_t* __restrict__ buckets, uint32_t* __restrict__ lane_partials, int bucket_r29) {
    constexpr int N = C::N, AV = N;                                 // 16-byte vectors per affine G2 table entry (4 N words)
    typedef typename Accum29G2<C>::Acc Acc;
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
    const Acc A{lds_acc29 + threadIdx.x};
    bool inf = true;
    // Software pipeline as in the G1 kernel: the next point (packed, AV x 16 bytes) is gathered while the current addition runs. On a box
    // with fast address translation this measures the same as the plain loop (the kernel is issue-bound); on boxes where random 128-byte
    // gathers over the 1.7 GB table are slow (r02: the same binary took 5.8 ms instead of 3.3 ms without it) two waves per SIMD cannot hide
    // the gather latency by themselves.
    const uint32_t s0 = starts[g];
    ListReader list(sorted, s0 + lo, s0 + hi);
    struct Raw { uint4 v[AV]; };
    auto fetch = [&](uint32_t& e_out, Raw& r_out) -> bool {
        while (list.more()) {
            const uint32_t e = list.next();
            uint32_t idx = e & 0x7fffffffu;
            if (idx < skip) continue;
            idx -= skip;
            if ((infmask[idx >> 5] >> (idx & 31)) & 1u) continue;
            const uint4* p = reinterpret_cast<const uint4*>(bases + (size_t)idx * (4 * N));
#pragma unroll
            for (int i = 0; i < AV; i++) r_out.v[i] = p[i];
            e_out = e;
            return true;
        }
        return false;
    };
    // the current point again, from the table (the rare doubling branch of the Jacobian form asks for it instead of keeping it live)
    auto regather = [&](uint32_t e, F2x<C>& qx, F2x<C>& qy) {
        const uint4* p = reinterpret_cast<const uint4*>(bases + (size_t)((e & 0x7fffffffu) - skip) * (4 * N));
        uint4 v[AV];
#pragma unroll
        for (int i = 0; i < AV; i++) v[i] = p[i];
        qx.c0 = unpack29_v<C>(v); qx.c1 = unpack29_v<C>(v + AV / 4); qy.c0 = unpack29_v<C>(v + AV / 2); qy.c1 = unpack29_v<C>(v + 3 * AV / 4);
        if (e >> 31) { qy.c0 = neg29<C, 2>(qy.c0); qy.c1 = neg29<C, 2>(qy.c1); }
    };
    if constexpr (ZK_G2_PREFETCH(C) && ZK_SPEC_GATHER_G2(C)) {
        auto fetch_spec = [&](SpecEntry& x, Raw& r_out) -> bool {
            if (!list.more()) return false;
            const uint32_t e = list.next();
            uint32_t idx = e & 0x7fffffffu;
            const bool mine = idx >= skip;
            idx = mine ? idx - skip : 0u;
            x.e = e; x.s = mine ? (idx & 31u) : 32u;
            x.m = infmask[idx >> 5];
            const uint4* p = reinterpret_cast<const uint4*>(bases + (size_t)idx * (4 * N));
#pragma unroll
            for (int i = 0; i < AV; i++) r_out.v[i] = p[i];
            return true;
        };
        SpecEntry x_next{0, 0, 32};
        Raw r_next;
        bool have = fetch_spec(x_next, r_next);
        while (have) {
            SpecEntry x = x_next;
            Raw r = r_next;
            have = fetch_spec(x_next, r_next);                      // the next bitmap word and gather are in flight during this addition
            while (spec_dead(x) && have) { x = x_next; r = r_next; have = fetch_spec(x_next, r_next); }
            if (spec_dead(x)) break;
            const uint32_t e = x.e;
            F2x<C> qx{unpack29_v<C>(r.v), unpack29_v<C>(r.v + AV / 4)}, qy{unpack29_v<C>(r.v + AV / 2), unpack29_v<C>(r.v + 3 * AV / 4)};
            if (e >> 31) { qy.c0 = neg29<C, 2>(qy.c0); qy.c1 = neg29<C, 2>(qy.c1); }      // 2p - y
            Accum29G2<C>::madd(A, inf, qx, qy, [&](F2x<C>& x2, F2x<C>& y2) { regather(e, x2, y2); });
        }
    } else if constexpr (ZK_G2_PREFETCH(C)) {
        uint32_t e_next = 0;
        Raw r_next;
        bool have = fetch(e_next, r_next);
        while (have) {
            const uint32_t e = e_next;
            const Raw r = r_next;
            have = fetch(e_next, r_next);                           // the next gather is in flight during this addition
            F2x<C> qx{unpack29_v<C>(r.v), unpack29_v<C>(r.v + AV / 4)}, qy{unpack29_v<C>(r.v + AV / 2), unpack29_v<C>(r.v + 3 * AV / 4)};
            if (e >> 31) { qy.c0 = neg29<C, 2>(qy.c0); qy.c1 = neg29<C, 2>(qy.c1); }      // 2p - y
            Accum29G2<C>::madd(A, inf, qx, qy, [&](F2x<C>& x2, F2x<C>& y2) { regather(e, x2, y2); });
        }
    } else {
        uint32_t e;
        Raw r;
        while (fetch(e, r)) {
            F2x<C> qx{unpack29_v<C>(r.v), unpack29_v<C>(r.v + AV / 4)}, qy{unpack29_v<C>(r.v + AV / 2), unpack29_v<C>(r.v + 3 * AV / 4)};
            if (e >> 31) { qy.c0 = neg29<C, 2>(qy.c0); qy.c1 = neg29<C, 2>(qy.c1); }
            Accum29G2<C>::madd(A, inf, qx, qy, [&](F2x<C>& x2, F2x<C>& y2) { regather(e, x2, y2); });
        }
    }
    // lane partials go to the trees in the reference's R-form; finished buckets stay in R'-form when the row / column sums run on the same limbs
    if (j) Accum29G2<C>::template store<false>(lane_partials + (size_t)lane * (8 * N), A, inf);
    else if (bucket_r29) Accum29G2<C>::template store<true>(buckets + (size_t)g * (8 * N), A, inf);
    else Accum29G2<C>::template store<false>(buckets + (size_t)g * (8 * N), A, inf);
}

// ---- r06: G2 accumulation with ONE Fq2 COMPONENT PER LANE (both curves) --------------------------------------------------------------------
// The two components of a bucket sit eight lanes apart in a row of sixteen (lanes 0-7 of a row: c0, lanes 8-15: c1). The accumulator is half as wide
// and stays in REGISTERS (151 VGPRs, three waves per SIMD, no LDS), and every Fq2 product fetches the partner's operands with DPP moves (row_ror:8; the
// move's bank mask does the per-component selection, no v_cndmask is spent): (a0 + a1 u)(b0 + b1 u) — the c0 lane forms a0 b0 + a1 (K p - b1), the c1
// lane a1 b0 + a0 b1, each ONE mul29_2 on the operands madd29_lds hands the same product, so every coordinate leaves with the same words. Measured first
// in isolation (tools/maddbench29_g2, profiles/r06_g2_layout.txt): 5.10 against 4.60 G additions/s for the LDS-parked layout. 14-limb moduli: 248 VGPRs, two
// waves per SIMD, no spill (the packed Jacobian in LDS: 256 + 111 spilled) and a hot loop of 61 KB instead of 118 KB; XYZZ there too, so the buckets are another
// representative of the same points than k_msm_accum29_g2's.
#if defined(__HIP_DEVICE_COMPILE__)
constexpr int DPP_ROR8 = 0x128;                                      // row_ror:8 — lane i of a row of 16 reads lane i ^ 8
// the partner's value in every lane
template <class C> ZK_DEV Fp29<C> xch_all(const Fp29<C>& v) {
    Fp29<C> r;
#pragma unroll
    for (int i = 0; i < Lim29<C>::NL; i++) {
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], DPP_ROR8, 0xF, 0xF, false);
        // keeps the move a move: folded into its only user (v_subrev_u32_dpp ... bound_ctrl:1 for the a0 - a1 of a square) the difference came out WRONG on
        // the device, while the unfolded form of the same subtraction was right (tools/maddbench29_g2 -DZK_DPP_FOLD shows it)
        asm volatile("" : "+v"(r.l[i]));
    }
    return r;
}
// lanes of the banks in BANKS (0x3: the c0 lanes, 0xC: the c1 lanes) take the PARTNER's src, the others keep `keep`
template <class C, int BANKS> ZK_DEV Fp29<C> xch_into(const Fp29<C>& keep, const Fp29<C>& src) {
    Fp29<C> r;
#pragma unroll
    for (int i = 0; i < Lim29<C>::NL; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)keep.l[i], (int)src.l[i], DPP_ROR8, 0xF, BANKS, false);
    return r;
}
// both lanes of a pair hold the AND of their flags
ZK_DEV bool pair_and(bool f) { const int z = f ? 1 : 0; return (z & __builtin_amdgcn_update_dpp(0, z, DPP_ROR8, 0xF, 0xF, false)) != 0; }
// the two right-hand operands of a split Fq2 product by b (own component bm, K p - bm in nb): y1 = b0 in both lanes, y2 = K p - b1 | b1
template <class C> struct BRole29 { Fp29<C> y1, y2; };
template <class C> ZK_DEV BRole29<C> b_role29(const Fp29<C>& bm, const Fp29<C>& nb) { return BRole29<C>{xch_into<C, 0xC>(bm, bm), xch_into<C, 0x3>(bm, nb)}; }
// own component of a * b: am / ao = own / partner's component of a
template <class C> ZK_DEV Fp29<C> f2mul_split(const Fp29<C>& am, const Fp29<C>& ao, const BRole29<C>& b) { return mul29_2(am, b.y1, ao, b.y2); }
// own component of a^2 (components <= KB): the c0 lane forms (a0 + a1)(a0 - a1), the c1 lane 2 a0 a1; c1mask = all ones in the c1 lanes
template <class C, int KB> ZK_DEV Fp29<C> f2sqr_split(const Fp29<C>& am, const Fp29<C>& ao, uint32_t c1mask) {
    const Fp29<C> u = add29(am, xch_into<C, 0x3>(zero29<C>(), am));  // c0 lanes: a0 + a1; c1 lanes: a1
    Fp29<C> d = sub29<C, KB>(am, ao); norm29(d);                      // c0 lanes: a0 - a1 (the c1 lanes' value is replaced)
    const Fp29<C> v = xch_into<C, 0xC>(d, am);                        // c1 lanes: a0
    Fp29<C> r = mul29(u, v);
#pragma unroll
    for (int i = 0; i < Lim29<C>::NL; i++) r.l[i] += r.l[i] & c1mask;
    norm29(r);
    return r;
}
template <class C> struct AccS29 { Fp29<C> X, Y, ZZ, ZZZ; };         // own component of every coordinate
// acc += q: madd29_lds line by line — same formulas, same offsets, the same operands per component (its invariants hold per component: both
// components of every value share one bound there). Both lanes of a pair take every branch together.
template <class C> ZK_DEV void madd29_split(AccS29<C>& a, bool& inf, const Fp29<C>& qx, const Fp29<C>& qy, uint32_t c1mask) {
    constexpr int NL = Lim29<C>::NL;
    if (inf) {
        a.X = qx; a.Y = qy;
        const Fp29<C> one = one29<C>();
#pragma unroll
        for (int i = 0; i < NL; i++) a.ZZ.l[i] = a.ZZZ.l[i] = one.l[i] & ~c1mask;                  // (1, 0)
        inf = false;
        return;
    }
    const Fp29<C> ZZo = xch_all(a.ZZ);
    const Fp29<C> U2 = f2mul_split(a.ZZ, ZZo, b_role29(qx, neg29<C, 2>(qx)));
    ZK_SFENCE();
    Fp29<C> P = sub29<C, 9>(U2, a.X); norm29(P);
    const Fp29<C> ZZZo = xch_all(a.ZZZ);
    const Fp29<C> S2 = f2mul_split(a.ZZZ, ZZZo, b_role29(qy, neg29<C, 3>(qy)));
    ZK_SFENCE();
    Fp29<C> R = sub29<C, 4>(S2, a.Y); norm29(R);
    if (pair_and(is_zero29(P))) {
        if (pair_and(is_zero29(R))) {
            // acc = 2 q (mdbl-2008-s-1, a = 0); rare: equal points in one bucket
            Fp29<C> U = add29(qy, qy); norm29(U);
            const Fp29<C> Uo = xch_all(U), qxo = xch_all(qx);
            const Fp29<C> V = f2sqr_split<C, 5>(U, Uo, c1mask);
            const BRole29<C> bV = b_role29(V, neg29<C, 3>(V));
            const Fp29<C> W = f2mul_split(U, Uo, bV), S = f2mul_split(qx, qxo, bV);
            const Fp29<C> xx = f2sqr_split<C, 2>(qx, qxo, c1mask);
            Fp29<C> M = add29(add29(xx, xx), xx); norm29(M);
            const Fp29<C> Mo = xch_all(M);
            Fp29<C> X3 = sub29<C, 2>(sub29<C, 2>(f2sqr_split<C, 7>(M, Mo, c1mask), S), S); norm29(X3);
            Fp29<C> Tt = sub29<C, 7>(S, X3); norm29(Tt);
            const Fp29<C> MT = f2mul_split(M, Mo, b_role29(Tt, neg29<C, 9>(Tt)));
            const Fp29<C> Wy = f2mul_split(W, xch_all(W), b_role29(qy, neg29<C, 3>(qy)));
            Fp29<C> Y3 = sub29<C, 2>(MT, Wy); norm29(Y3);
            a.X = X3; a.Y = Y3; a.ZZ = V; a.ZZZ = W;
        } else inf = true;
        return;
    }
    const Fp29<C> Po = xch_all(P);
    const Fp29<C> PP = f2sqr_split<C, 11>(P, Po, c1mask);
    ZK_SFENCE();
    const BRole29<C> bPP = b_role29(PP, neg29<C, 4>(PP));
    const Fp29<C> PPP = f2mul_split(P, Po, bPP);
    ZK_SFENCE();
    const Fp29<C> nPPP = neg29<C, 2>(PPP);
    const BRole29<C> bPPP = b_role29(PPP, nPPP);
    a.ZZ = f2mul_split(a.ZZ, ZZo, bPP);
    ZK_SFENCE();
    a.ZZZ = f2mul_split(a.ZZZ, ZZZo, bPPP);
    ZK_SFENCE();
    const Fp29<C> Q = f2mul_split(a.X, xch_all(a.X), bPP);
    ZK_SFENCE();
    const Fp29<C> RR = f2sqr_split<C, 6>(R, xch_all(R), c1mask);
    Fp29<C> X3 = sub29<C, 2>(sub29<C, 2>(sub29<C, 2>(RR, PPP), Q), Q); norm29(X3);
    a.X = X3;
    ZK_SFENCE();
    Fp29<C> Tq = sub29<C, 9>(Q, X3); norm29(Tq);
    // Y3 = Tq R + Y1 (-PPP), one reduction per component: the right-hand operands of R and of N = (2p - PPP0, 2p - PPP1)
    const BRole29<C> bR = b_role29(R, neg29<C, 6>(R));
    const BRole29<C> bN{xch_into<C, 0xC>(nPPP, nPPP), xch_into<C, 0x3>(nPPP, PPP)};
    if constexpr (ZK_Y3_SPLIT(C)) {                                 // 14-limb moduli: two 2-product sums instead of one 4-product sum (half the live operands), as in madd29_lds
        const Fp29<C> TR = mul29_2(Tq, bR.y1, xch_all(Tq), bR.y2);
        ZK_SFENCE();
        Fp29<C> Y3 = add29(TR, mul29_2(a.Y, bN.y1, xch_all(a.Y), bN.y2));
        norm29(Y3);
        a.Y = Y3;
    } else a.Y = mul29_4(Tq, bR.y1, xch_all(Tq), bR.y2, a.Y, bN.y1, xch_all(a.Y), bN.y2);
}
#endif
// 128 schedule lanes (buckets or shares of a bucket) per 256-thread block; same arguments and results as k_msm_accum29_g2
constexpr unsigned G2S_SLOTS = 128;
template <class C> __global__ void __launch_bounds__(256, Lim29<C>::NL <= 9 ? 3 : 2)
k_msm_accum29_g2s(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ infmask, MsmShape sh, uint32_t skip, uint32_t cap, const uint32_t* __restrict__ counts,
                  const uint32_t* __restrict__ starts, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ lane_g, const uint32_t* __restrict__ lane_sub,
                  const uint32_t* __restrict__ meta, uint32_t* __restrict__ buckets, uint32_t* __restrict__ lane_partials, int bucket_r29) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int N = C::N, HV = N / 4;                              // 16-byte vectors per component of a coordinate
    const uint32_t t = threadIdx.x, comp = (t >> 3) & 1u, c1mask = comp ? 0xffffffffu : 0u;
    const uint32_t lane = blockIdx.x * G2S_SLOTS + (((t >> 4) << 3) | (t & 7u));      // the schedule lane this PAIR works for
    if (lane >= meta[0]) return;                                     // the partner leaves with it
    const uint32_t g = lane_g[lane];
    const uint32_t cnt = counts[g];
    const uint32_t j = msm_key_lanes_log(msm_key(cnt, cap), cap);
    uint32_t lo = 0, hi = cnt;
    if (j) {
        const uint32_t chunk = (cnt + (1u << j) - 1) >> j;
        lo = min(cnt, lane_sub[lane] * chunk);
        hi = min(cnt, lo + chunk);
    }
    AccS29<C> a;
    bool inf = true;
    const uint32_t s0 = starts[g];
    ListReader list(sorted, s0 + lo, s0 + hi);
    struct Raw { uint4 v[2 * HV]; };                                 // own component of x, own component of y
    auto fetch = [&](uint32_t& e_out, Raw& r_out) -> bool {
        while (list.more()) {
            const uint32_t e = list.next();
            uint32_t idx = e & 0x7fffffffu;
            if (idx < skip) continue;
            idx -= skip;
            if ((infmask[idx >> 5] >> (idx & 31)) & 1u) continue;
            const uint4* p = reinterpret_cast<const uint4*>(bases + (size_t)idx * (4 * N) + comp * N);
#pragma unroll
            for (int i = 0; i < HV; i++) { r_out.v[i] = p[i]; r_out.v[HV + i] = p[2 * HV + i]; }
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
        have = fetch(e_next, r_next);                               // the next gather is in flight during this addition
        const Fp29<C> qx = unpack29_v<C>(r.v);
        Fp29<C> qy = unpack29_v<C>(r.v + HV);
        if (e >> 31) qy = neg29<C, 2>(qy);                          // 2p - y
        madd29_split<C>(a, inf, qx, qy, c1mask);
    }
    // own component of every coordinate: word offset (2 coordinate + component) N of the point
    uint32_t* dst = (j ? lane_partials + (size_t)lane * (8 * N) : buckets + (size_t)g * (8 * N)) + comp * N;
    const bool keep29 = !j && bucket_r29;
    if (inf) {
#pragma unroll
        for (int cdn = 0; cdn < 4; cdn++)
#pragma unroll
            for (int i = 0; i < HV; i++) reinterpret_cast<uint4*>(dst + cdn * 2 * N)[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    if (keep29) { store_r256<C, true>(dst, a.X); store_r256<C, true>(dst + 2 * N, a.Y); store_r256<C, true>(dst + 4 * N, a.ZZ); store_r256<C, true>(dst + 6 * N, a.ZZZ); }
    else { store_r256<C, false>(dst, a.X); store_r256<C, false>(dst + 2 * N, a.Y); store_r256<C, false>(dst + 4 * N, a.ZZ); store_r256<C, false>(dst + 6 * N, a.ZZZ); }
#endif
}

// ---- row / column sums of the Fq2 bucket reduction on unsaturated limbs ------------------------------------------------------------------
// acc (XYZZ over Fq2, parked in LDS) = 2 acc (dbl-2008-s-1, a = 0): the rare equal-points branch of padd29_lds
template <class C, class Acc> ZK_HD void dbl29_lds(const Acc& A) {
    F2x<C> t;
    A.get(1, t);
    F2x<C> U{add29(t.c0, t.c0), add29(t.c1, t.c1)}; f2norm(U);                          // Y <= 3.8: U <= 7.6
    const F2x<C> V = f2sqr<C, 8>(U);                                                    // <= 2.5
    const Fp29<C> nV1 = neg29<C, 3>(V.c1);
    const F2x<C> W = f2mul(U, V, nV1);                                                  // <= 1.3
    const Fp29<C> nY0 = neg29<C, 4>(t.c0), nY1 = neg29<C, 4>(t.c1);
    const F2x<C> nWY{mul29_2(W.c0, nY0, W.c1, t.c1), mul29_2(W.c0, nY1, W.c1, nY0)};    // - W Y1 <= 1.1
    A.get(0, t);
    const F2x<C> S = f2mul(t, V, nV1);                                                  // X1 V <= 1.3
    const F2x<C> xx = f2sqr<C, 9>(t);                                                   // <= 2.9
    F2x<C> M{add29(add29(xx.c0, xx.c0), xx.c0), add29(add29(xx.c1, xx.c1), xx.c1)}; f2norm(M);      // <= 8.6
    F2x<C> X3 = f2sub<C, 2>(f2sub<C, 2>(f2sqr<C, 9>(M), S), S); f2norm(X3);             // <= 2.9 + 4 = 6.9
    A.put(0, X3);
    F2x<C> Tt = f2sub<C, 8>(S, X3); f2norm(Tt);                                         // <= 9.3
    const F2x<C> MT = f2mul(M, Tt, neg29<C, 10>(Tt.c1));                                // <= 2
    F2x<C> Y3{add29(MT.c0, nWY.c0), add29(MT.c1, nWY.c1)}; f2norm(Y3);                  // <= 3.1
    A.put(1, Y3);
    A.get(2, t); A.put(2, f2mul(t, V, nV1));
    A.get(3, t); A.put(3, f2mul(t, W, neg29<C, 2>(W.c1)));
}
// acc (XYZZ over Fq2, parked in LDS) += p for a general XYZZ point whose coordinates are fetched on demand: ld(k, F2x&), k = 0..3 = X, Y, ZZ,
// ZZZ (add-2008-s, 12M + 2S in Fq2). Invariants of madd29_lds for the accumulator (X <= 8.4, Y <= 3.8, ZZ, ZZZ <= 1.1, normalised); p within
// the same bounds (a point just unpacked from R'-form words is canonical).
template <class C, class Acc, class Ld> ZK_HD void padd29_lds(const Acc& A, bool& inf, Ld ld) {
    F2x<C> t, u;
    if (inf) {
#pragma unroll 1
        for (int k = 0; k < 4; k++) { ld(k, u); A.put(k, u); }
        inf = false;
        return;
    }
    // Ordered for few live values (the 14-limb instance holds 28 registers per Fq2 element): S1 = Y1 ZZZ2 is formed twice — once for R, once
    // for Y3 at the very end, while Y1 still sits in its slot — instead of being carried across the whole addition (one Fq2 product of 15).
    ld(2, u); A.get(0, t);
    const F2x<C> U1 = f2mul(t, u, neg29<C, 2>(u.c1));                                   // X1 ZZ2 <= 1.2
    ZK_SFENCE();
    ld(0, u); A.get(2, t);
    F2x<C> P = f2sub<C, 2>(f2mul(u, t, neg29<C, 2>(t.c1)), U1); f2norm(P);              // X2 ZZ1 - U1 <= 3.2
    ZK_SFENCE();
    ld(3, u); A.get(1, t);
    F2x<C> R = f2mul(t, u, neg29<C, 2>(u.c1));                                          // S1 = Y1 ZZZ2 <= 1.1
    ZK_SFENCE();
    ld(1, u); A.get(3, t);
    R = f2sub<C, 2>(f2mul(u, t, neg29<C, 2>(t.c1)), R); f2norm(R);                      // Y2 ZZZ1 - S1 <= 3.2
    ZK_SFENCE();
    if (f2zero(P)) {
        if (f2zero(R)) dbl29_lds<C>(A); else inf = true;
        return;
    }
    const F2x<C> PP = f2sqr<C, 4>(P);                                                   // c0 <= 1.3, c1 = 2 a0 a1 <= 2.2
    ZK_SFENCE();
    const Fp29<C> nPP1 = neg29<C, 3>(PP.c1);
    const F2x<C> Q = f2mul(U1, PP, nPP1);                                               // <= 1.1
    ZK_SFENCE();
    const F2x<C> PPP = f2mul(P, PP, nPP1);                                              // <= 1.1
    ZK_SFENCE();
    ld(2, u); A.get(2, t);
    t = f2mul(t, u, neg29<C, 2>(u.c1));
    ZK_SFENCE();
    A.put(2, f2mul(t, PP, nPP1));                                                       // ZZ3 = ZZ1 ZZ2 PP
    ZK_SFENCE();
    const Fp29<C> nPPP1 = neg29<C, 2>(PPP.c1);
    ld(3, u); A.get(3, t);
    t = f2mul(t, u, neg29<C, 2>(u.c1));
    ZK_SFENCE();
    A.put(3, f2mul(t, PPP, nPPP1));                                                     // ZZZ3 = ZZZ1 ZZZ2 PPP
    ZK_SFENCE();
    F2x<C> Tq = f2sub<C, 2>(f2sub<C, 2>(f2sub<C, 2>(f2sqr<C, 4>(R), PPP), Q), Q); f2norm(Tq);     // X3 <= 2.2 + 6 = 8.2 (the square's c1 is a doubled product)
    A.put(0, Tq);
    ZK_SFENCE();
    Tq = f2sub<C, 9>(Q, Tq); f2norm(Tq);                                                // Q - X3 <= 10.1
    const Fp29<C> nR1 = neg29<C, 4>(R.c1);
    const F2x<C> TR{mul29_2(Tq.c0, R.c0, Tq.c1, nR1), mul29_2(Tq.c0, R.c1, Tq.c1, R.c0)};         // <= (32 + 40) / 169 + 1 = 1.5
    ZK_SFENCE();
    ld(3, u); A.get(1, t);
    const F2x<C> S1 = f2mul(t, u, neg29<C, 2>(u.c1));                                   // again: Y1 ZZZ2 (ld(3) is the OPERAND's ZZZ, untouched)
    ZK_SFENCE();
    const Fp29<C> nPPP0 = neg29<C, 2>(PPP.c0);
    F2x<C> Y3;
    Y3.c0 = add29(TR.c0, mul29_2(S1.c0, nPPP0, S1.c1, PPP.c1));                         // - S1 PPP: <= 1.4 + 1.1
    Y3.c1 = add29(TR.c1, mul29_2(S1.c0, nPPP1, S1.c1, nPPP0));
    f2norm(Y3);
    A.put(1, Y3);
}
// XYZZ accumulators for the reduction kernels: the accumulation's lane layout, four coordinates
template <class C> struct Reduce29G2 {
    static constexpr int T = MsmAccumBlock<Fp2<C>>::value;
    static constexpr bool PACK = C::N > 8;
    typedef LdsAcc29<C, T, PACK> Acc;
    static constexpr size_t lds_bytes = (size_t)T * 8 * Acc::EW * 4;
};
// one wave per row / column sum over R'-form Fq2 buckets (see k_msm_rowcol_wave29); the sums leave in the reference's R-form
template <class C> __global__ void __launch_bounds__(Reduce29G2<C>::T, 2)
k_msm_rowcol_wave29_g2(MsmReduceBatch rb, uint32_t W, uint32_t nb, uint32_t rbits, uint32_t cbits, uint32_t* __restrict__ out) {
    constexpr int N = C::N, PW = 8 * N, T = Reduce29G2<C>::T;
    typedef typename Reduce29G2<C>::Acc Acc;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t inf_s[T];
    const uint32_t Cn = 1u << cbits, R = 1u << rbits;
    const uint32_t t = threadIdx.x, sub = t & 63u;
    const size_t n_out = (size_t)rb.njobs * W * 2 * Cn;
    const Acc A{lds + t};
    for (size_t blk = blockIdx.x; blk * (T / 64) < n_out; blk += gridDim.x) {
        const size_t gw = blk * (T / 64) + (t >> 6);             // sum index: ((job*W + w)*2 + kind)*C + i
        const bool valid = gw < n_out;
        const uint32_t i = (uint32_t)(gw & (Cn - 1)), kind = (uint32_t)(gw >> cbits) & 1u;
        const size_t jw = gw >> (cbits + 1);
        const uint32_t w = (uint32_t)(jw % W), job = valid ? (uint32_t)(jw / W) : 0u;
        const uint32_t* bk = rb.buckets[job];
        const uint32_t* cn = rb.counts[job];
        const uint32_t cnt = !valid ? 0u : (kind ? R : ((i < R) ? Cn : 0u));
        bool inf = true;
        for (uint32_t e = sub; e < cnt; e += 64) {
            const size_t g = (size_t)w * nb + (kind ? ((size_t)e << cbits) + i : ((size_t)i << cbits) + e);
            const uint32_t* src = bk + g * PW;
            if (cn[g] && !xyzz29_words_inf_g2<C>(src))
                padd29_lds<C>(A, inf, [&](int k, F2x<C>& v) { v.c0 = load29_packed<C>(src + k * 2 * N); v.c1 = load29_packed<C>(src + k * 2 * N + N); });
        }
        for (uint32_t d = 1; d < 64; d <<= 1) {
            inf_s[t] = inf ? 1u : 0u;
            __syncthreads();
            if ((t & (2 * d - 1)) == 0 && !inf_s[t + d]) {
                const Acc Pn{A.base + d};
                padd29_lds<C>(A, inf, [&](int k, F2x<C>& v) { Pn.get(k, v); });
            }
            __syncthreads();
        }
        if (valid && sub == 0) store_xyzz29_lds<C, Acc, false>(out + gw * PW, A, inf);
        __syncthreads();
    }
}

}  // namespace zkmi
