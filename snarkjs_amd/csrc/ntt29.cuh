// snarkjs_amd/csrc/ntt29.cuh — the Fr NTT passes of ntt.cuh on unsaturated 9 x 29-bit limbs (field29.cuh), gfx950.
//
// Same passes, tiles, index algebra and launch geometry as ntt.cuh (read its header first); what changes is the arithmetic inside a tile:
//   * the data keeps the reference's Montgomery factor: an element x is carried as the VALUE v = x 2^256 (what the caller's bytes hold), as 9
//     lazy limbs. Twiddles, row factors, pre-scale factors and 1/n are stored in R'-form (t 2^261 mod r, packed canonical words), so that
//     mul29(v, t') = v t 2^261 / 2^261 = (x t) 2^256: the product is again a value in the caller's form — no conversion anywhere;
//   * butterflies are decimation-in-TIME: (x, y) -> (x + w y, x - w y + 2r). The multiplication comes first, so a lazy sum never meets
//     another lazy sum multiplicatively: values grow by at most 2r per stage (< 1.3r + 2r l <= 19.3r < 2^258 after l <= 9 stages, r02's
//     decimation-in-frequency form would double them per stage). A product is 207 instructions against ~290 on saturated limbs, and the
//     addition / subtraction are 9 limb operations + one carry pass each instead of two carry chains with conditional corrections;
//   * the DIT form used here takes its input in NATURAL order and leaves the output bit-reversed, exactly like the DIF stages of ntt.cuh, so
//     the loads, the transposed stores and the digit reversal of the last pass are unchanged; stage s (span h = 2^s, from l-1 down to 0)
//     multiplies the upper half of block `blk` by w_N^(bitrev(blk) 2^s) = U[blk] with U the local twiddle table in bit-reversed order:
//     lanes read consecutive or identical entries (no bank conflicts, like the jl << (l-1-s) indexing of the DIF form);
//   * between passes the work array holds the lazy values as 12-word records (9 limbs + padding: 48 bytes per element, three 16-byte
//     vector accesses per lane and whole 64-byte sectors per group of four columns — 36-byte records cost nine scattered dword accesses per
//     lane and ran the middle passes 8x slower); only the last pass brings
//     every value to the canonical range (reduce29_small: quotient estimate from the top limb, one multiply-subtract pass, <= 3 conditional
//     subtractions) and writes the reference's 32 bytes.
#pragma once
#include "field29.cuh"
#include "ntt.cuh"

namespace zkmi {

constexpr int NTT29_REC = 12;                 // words per element of the work array between passes (9 limbs + 3 words of padding)
// r05 measured two restructurings of the stage loop against the shipped one (profiles/NOTES.md "Round 5", profiles/r05_ntt_ab.txt; both bit-identical,
// neither faster). They stay buildable for the A/B (ZKMI_BUILD_VARIANT=<v> ZKMI_EXTRA_FLAGS=-DZKMI_NTT_VARIANT=<k>, tools/lab/r5_ntt_ab.sh):
//   0  shipped: row-major tile in the strided passes, one block barrier per stage
//   1  transposed tile in the strided passes too; a column is worked on by ONE wave in every stage, the barrier between stages becomes a
//      wavefront-scope fence (one block barrier after the last stage)                                             2^20: 0.1497 vs 0.1505 ms
//   2  = 1 with radix-4 groups held in registers (two stages per LDS round trip), 128-thread blocks                2^20: 0.187 ms
#ifndef ZKMI_NTT_VARIANT
#define ZKMI_NTT_VARIANT 0
#endif
constexpr int NTT29_THREADS = ZKMI_NTT_VARIANT == 2 ? 128 : NTT_THREADS;

// v (normalised, < 32 r) -> v mod r, canonical limbs
template <class C> ZK_HD void reduce29_small(Fp29<C>& v) {
    using L = Lim29<C>;
    constexpr int NL = L::NL, B = L::B;
    static_assert(NL == 9 && B == 29, "Fr form");
#if defined(ZK29_SHADOW)
    b29::need(b29::normalised(v) && v.bv <= 32.0, "reduce29_small: operand not normalised or not below 32 r", v.bv, 32.0);
    b29::set(v, 1.0, b29::lowmax<C>(), -1.0);
#endif
    // q' <= floor(v / r) <= q' + 2: floor(top limb / (top limb of r + 1)) by a 40-bit reciprocal
    constexpr uint64_t MAGIC = (1ull << 40) / ((uint64_t)L::p(NL - 1) + 1);
    const uint32_t q = (uint32_t)(((uint64_t)v.l[NL - 1] * MAGIC) >> 40);
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int64_t t = (int64_t)v.l[i] - (int64_t)((uint64_t)q * L::p(i)) + c;
        if (i < NL - 1) { v.l[i] = (uint32_t)t & mask29<C>(); c = t >> B; } else v.l[i] = (uint32_t)t;
    }
    // < 3 r left: subtract r while >= r
#pragma unroll 1
    for (int rep = 0; rep < 3; rep++) {
        int32_t d[NL], cc = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) { int32_t t = (int32_t)v.l[i] - (int32_t)L::p(i) + cc; if (i < NL - 1) { cc = t >> B; d[i] = t & (int32_t)mask29<C>(); } else d[i] = t; }
        const bool ge = d[NL - 1] >= 0;
#pragma unroll
        for (int i = 0; i < NL; i++) v.l[i] = ge ? (uint32_t)d[i] : v.l[i];
    }
}
// one DIT butterfly: (x, y) -> (x + t, x - t + 2r), t = w y (or y itself in the first stage, where w = 1); x lazy (< 2^258, normalised),
// t normalised and < 2r - 2^232; both results normalised
template <class C> ZK_HD void ntt29_bfly(Fp29<C>& x, Fp29<C>& y, const Fp29<C>& t) {
    const Fp29<C> s = add29(x, t);
    y = sub29<C, 2>(x, t);
    x = s;
    norm29(x); norm29(y);
}

// Tile storage in LDS (r04): THREE planes per region — limbs 0..3 of element e as one 16-byte slot of plane a, limbs 4..7 of plane b, limb 8 as
// one word of plane c — so that an element moves with 2 x ds_*_b128 + 1 x ds_*_b32 instead of the nine 4-byte accesses of r03's limb planes
// (45 LDS instructions per butterfly, which ate what the cheaper product saved; now 15). Consecutive lanes touch consecutive slots: 16
// consecutive lanes of a b128 access cover all 64 banks once, like the two 16-byte planes of ntt.cuh. Regions are padded to whole groups of
// four elements so that every plane starts 16-byte aligned.
ZK_HD constexpr uint32_t ntt29_al4(uint32_t n) { return (n + 3u) & ~3u; }
struct Tile29 {
    uint4 *a, *b;
    uint32_t* c;
    ZK_DEV Tile29(uint32_t* base, uint32_t n_elems) : a(reinterpret_cast<uint4*>(base)), b(a + ntt29_al4(n_elems)), c(reinterpret_cast<uint32_t*>(b + ntt29_al4(n_elems))) {}
    __host__ __device__ static constexpr uint32_t words(uint32_t n_elems) { return 9u * ntt29_al4(n_elems); }
};
template <class C> ZK_DEV Fp29<C> lds29_get(const Tile29& t, uint32_t e) {
    const uint4 x = t.a[e], y = t.b[e];
    Fp29<C> r;
    r.l[0] = x.x; r.l[1] = x.y; r.l[2] = x.z; r.l[3] = x.w; r.l[4] = y.x; r.l[5] = y.y; r.l[6] = y.z; r.l[7] = y.w; r.l[8] = t.c[e];
    return r;
}
template <class C> ZK_DEV void lds29_put(const Tile29& t, uint32_t e, const Fp29<C>& v) {
    t.a[e] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    t.b[e] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    t.c[e] = v.l[8];
}
// element `i` of an array of 48-byte records (the work array between passes)
template <class C> ZK_DEV Fp29<C> rec29_load(const uint32_t* base, uint64_t i) {
    const uint4* p = reinterpret_cast<const uint4*>(base + i * NTT29_REC);
    const uint4 a = p[0], b = p[1], c = p[2];
    Fp29<C> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
    return r;
}
template <class C> ZK_DEV void rec29_store(uint32_t* base, uint64_t i, const Fp29<C>& v) {
    uint4* p = reinterpret_cast<uint4*>(base + i * NTT29_REC);
    p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]); p[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]); p[2] = make_uint4(v.l[8], 0u, 0u, 0u);
}
// w^e from the split power tables (R'-form packed words): one product, normalised, < 1.1 r
template <class C> ZK_DEV Fp29<C> ntt29_pow(const NttPassArgs& a, uint64_t e) {
    return mul29(load29_packed<C>(a.T_lo + (size_t)(e & ((1ull << a.log_lb) - 1)) * C::N), load29_packed<C>(a.T_hi + (size_t)(e >> a.log_lb) * C::N));
}

// radix-2 DIT stages on an LDS tile, natural order in, bit-reversed order out. ROWMAJOR: element (row j, col c) at j*CH + c (strided
// passes); otherwise at c*(N+1) + j (last pass; +1 pad keeps the transposed store conflict-free). U: local twiddles in bit-reversed order.
template <class C, bool ROWMAJOR> ZK_DEV void ntt29_tile_stages(const Tile29& p, const Tile29& U, uint32_t l, uint32_t log_ch) {
    const uint32_t half_elems = 1u << (l + log_ch - 1);
    const uint32_t N = 1u << l, HN = N >> 1;
    // variant 1: with at most 64 pairs per column, butterfly b = (column b >> (l-1), pair b & (N/2-1)) of lane b mod 256 keeps every column inside one
    // wave through all stages; LDS operations of one wave execute in program order
    const bool wave_private = ZKMI_NTT_VARIANT >= 1 && !ROWMAJOR && HN <= 64u && (NTT29_THREADS % 64) == 0;       // block-uniform
    for (int s = (int)l - 1; s >= 0; s--) {
        const uint32_t h = 1u << s;
        for (uint32_t b = threadIdx.x; b < half_elems; b += NTT29_THREADS) {
            uint32_t c, pr;
            if (ROWMAJOR) { c = b & ((1u << log_ch) - 1); pr = b >> log_ch; }
            else { pr = b & (HN - 1); c = b >> (l - 1); }
            const uint32_t jl = pr & (h - 1), blk = pr >> s;
            const uint32_t j = (blk << (s + 1)) | jl;
            const uint32_t e0 = ROWMAJOR ? (j << log_ch) + c : c * (N + 1) + j;
            const uint32_t e1 = ROWMAJOR ? ((j + h) << log_ch) + c : c * (N + 1) + j + h;
            Fp29<C> x = lds29_get<C>(p, e0), y = lds29_get<C>(p, e1);
            if (s < (int)l - 1) y = mul29(y, lds29_get<C>(U, blk));            // first stage: one block, w = 1, the operand is a fresh product or canonical
            ntt29_bfly(x, y, Fp29<C>(y));
            lds29_put<C>(p, e0, x);
            lds29_put<C>(p, e1, y);
        }
        if (wave_private) {
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
        } else __syncthreads();
    }
    if (wave_private && l) __syncthreads();
}

#if ZKMI_NTT_VARIANT == 2
// variant 2: the same stages two at a time (transposed layout): a lane holds the four elements j, j + h/2, j + h, j + 3h/2 of stages s and s - 1
// (h = 2^s) in registers — stage s pairs (x0, x2), (x1, x3) under U[blk], stage s - 1 pairs (x0, x1) under U[2 blk] and (x2, x3) under U[2 blk + 1] —
// and touches LDS once per TWO stages. Element for element the operations (product, then lazy butterfly) are those of the radix-2 loop, so the
// values and their proven bounds are the same; a last single stage remains when l is odd, on the SAME groups (a lane takes the two adjacent pairs
// 4 pr .. 4 pr + 3 of its column) so that a column stays with the wave that ran its radix-4 steps.
template <class C> ZK_DEV void ntt29_tile_stages_r4(const Tile29& p, const Tile29& U, uint32_t l, uint32_t log_ch) {
    const uint32_t N = 1u << l, QN = N >> 2;
    const bool wave_private = (N >> 1) <= 64u && (NTT29_THREADS % 64) == 0;
    auto step_sync = [&]() {
        if (wave_private) {
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
        } else __syncthreads();
    };
    int s = (int)l - 1;
    if (l >= 2) {
        const uint32_t quarter_elems = 1u << (l + log_ch - 2);
        for (; s >= 1; s -= 2) {
            const uint32_t h = 1u << s, h2 = h >> 1;
            for (uint32_t q = threadIdx.x; q < quarter_elems; q += NTT29_THREADS) {
                const uint32_t pr = q & (QN - 1), c = q >> (l - 2);
                const uint32_t jl = pr & (h2 - 1), blk = pr >> (s - 1);
                const uint32_t e = c * (N + 1) + ((blk << (s + 1)) | jl);
                Fp29<C> x0 = lds29_get<C>(p, e), x1 = lds29_get<C>(p, e + h2), x2 = lds29_get<C>(p, e + h), x3 = lds29_get<C>(p, e + h + h2);
                if (s < (int)l - 1) { const Fp29<C> w = lds29_get<C>(U, blk); x2 = mul29(x2, w); x3 = mul29(x3, w); }
                ntt29_bfly(x0, x2, Fp29<C>(x2));
                ntt29_bfly(x1, x3, Fp29<C>(x3));
                x1 = mul29(x1, lds29_get<C>(U, 2 * blk));
                x3 = mul29(x3, lds29_get<C>(U, 2 * blk + 1));
                ntt29_bfly(x0, x1, Fp29<C>(x1));
                ntt29_bfly(x2, x3, Fp29<C>(x3));
                lds29_put<C>(p, e, x0); lds29_put<C>(p, e + h2, x1); lds29_put<C>(p, e + h, x2); lds29_put<C>(p, e + h + h2, x3);
            }
            step_sync();
        }
    }
    if (s == 0 && l >= 3) {
        const uint32_t quarter_elems = 1u << (l + log_ch - 2);
        for (uint32_t q = threadIdx.x; q < quarter_elems; q += NTT29_THREADS) {
            const uint32_t pr = q & (QN - 1), c = q >> (l - 2);
            const uint32_t e = c * (N + 1) + (pr << 2);
            Fp29<C> x0 = lds29_get<C>(p, e), x1 = lds29_get<C>(p, e + 1), x2 = lds29_get<C>(p, e + 2), x3 = lds29_get<C>(p, e + 3);
            x1 = mul29(x1, lds29_get<C>(U, 2 * pr));
            x3 = mul29(x3, lds29_get<C>(U, 2 * pr + 1));
            ntt29_bfly(x0, x1, Fp29<C>(x1));
            ntt29_bfly(x2, x3, Fp29<C>(x3));
            lds29_put<C>(p, e, x0); lds29_put<C>(p, e + 1, x1); lds29_put<C>(p, e + 2, x2); lds29_put<C>(p, e + 3, x3);
        }
        step_sync();
    } else if (s == 0) {                                                   // l = 1: one butterfly per column, w = 1
        const uint32_t half_elems = 1u << (l + log_ch - 1);
        for (uint32_t b = threadIdx.x; b < half_elems; b += NTT29_THREADS) {
            const uint32_t e0 = b * (N + 1);
            Fp29<C> x = lds29_get<C>(p, e0), y = lds29_get<C>(p, e0 + 1);
            ntt29_bfly(x, y, Fp29<C>(y));
            lds29_put<C>(p, e0, x);
            lds29_put<C>(p, e0 + 1, y);
        }
        step_sync();
    }
    if (wave_private && l) __syncthreads();
}
#endif
// the stage loop of a transposed tile in the variant this build selects
template <class C> ZK_DEV void ntt29_stages_t(const Tile29& p, const Tile29& U, uint32_t l, uint32_t log_ch) {
#if ZKMI_NTT_VARIANT == 2
    ntt29_tile_stages_r4<C>(p, U, l, log_ch);
#else
    ntt29_tile_stages<C, false>(p, U, l, log_ch);
#endif
}

// IN_REC: the input array holds 48-byte lazy records (the work array; strided passes always write it) instead of the caller's canonical 32-byte elements
// ---- passes 1 .. p-1 (in place over the FFT digit, columns contiguous in memory) ------------------------------------
template <class C, bool IN_REC> __global__ void __launch_bounds__(NTT29_THREADS)
k_ntt29_pass_strided(const uint32_t* in, uint32_t* out, NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds29[];
    in += (size_t)blockIdx.y * a.in_bs; out += (size_t)blockIdx.y * a.out_bs;
    const uint32_t l = a.l[a.pass], N = 1u << l, CH = 1u << a.log_ch, E = N << a.log_ch, HN = N >> 1;
    constexpr bool TR = ZKMI_NTT_VARIANT >= 1;                            // transposed tile: element (row j, column c) at c (N+1) + j instead of j CH + c
    const uint32_t PL = TR ? (N + 1) << a.log_ch : E;
    const Tile29 p(lds29, PL);                                            // the tile
    const Tile29 U(lds29 + Tile29::words(PL), HN);                        // N/2 local twiddles (bit-reversed order)
    const Tile29 rf(lds29 + Tile29::words(PL) + Tile29::words(HN), N);    // N row factors
    uint32_t log_S = 0;
    for (uint32_t m = a.pass + 1; m < a.n_pass; m++) log_S += a.l[m];
    const uint64_t tiles_per_u = (1ull << log_S) >> a.log_ch;
    const uint64_t u = blockIdx.x / tiles_per_u, q0 = (blockIdx.x % tiles_per_u) << a.log_ch;
    const uint64_t base = (u << (l + log_S)) + q0;
    const bool has_fac = (a.pass > 0) || (a.rowinc != nullptr);
    if (has_fac) {
        const uint64_t K = a.pass > 0 ? ntt_digit_reverse(a, u, 0, a.pass) : 0;
        for (uint32_t j = threadIdx.x; j < N; j += NTT29_THREADS) {
            Fp29<C> f;
            if (a.pass > 0) {
                const uint64_t e = (((uint64_t)j * K) << log_S) & ((1ull << a.log_n) - 1);
                f = ntt29_pow<C>(a, e);
                if (a.rowinc) f = mul29(f, load29_packed<C>(a.rowinc + (size_t)j * C::N));
            } else f = load29_packed<C>(a.rowinc + (size_t)j * C::N);
            lds29_put<C>(rf, j, f);
        }
    }
    for (uint32_t k = threadIdx.x; k < HN; k += NTT29_THREADS) lds29_put<C>(U, k, load29_packed<C>(a.LT + (size_t)k * C::N));
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < E; idx += NTT29_THREADS) {
        const uint32_t c = idx & (CH - 1), j = idx >> a.log_ch;
        const uint64_t src = base + ((uint64_t)j << log_S) + c;
        Fp29<C> x = IN_REC ? rec29_load<C>(in, src) : (a.in_len && src >= a.in_len) ? zero29<C>() : load29_packed<C>(in + src * C::N);
        if (has_fac) x = mul29(x, lds29_get<C>(rf, j));
        lds29_put<C>(p, TR ? c * (N + 1) + j : idx, x);
    }
    __syncthreads();
    if constexpr (TR) ntt29_stages_t<C>(p, U, l, a.log_ch); else ntt29_tile_stages<C, true>(p, U, l, a.log_ch);
    for (uint32_t idx = threadIdx.x; idx < E; idx += NTT29_THREADS) {
        const uint32_t c = idx & (CH - 1), k = idx >> a.log_ch;
        const uint32_t j = __brev(k) >> (32 - l);
        rec29_store<C>(out, base + ((uint64_t)k << log_S) + c, lds29_get<C>(p, TR ? c * (N + 1) + j : (j << a.log_ch) + c));
    }
}

// ---- last pass: contiguous j_p runs in, natural order out, canonical bytes ------------------------------------------------
template <class C, bool IN_REC> __global__ void __launch_bounds__(NTT29_THREADS)
k_ntt29_pass_last(const uint32_t* in, uint32_t* out, NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds29[];
    in += (size_t)blockIdx.y * a.in_bs; out += (size_t)blockIdx.y * a.out_bs;
    const uint32_t l = a.l[a.pass], N = 1u << l, CH = 1u << a.log_ch, E = N << a.log_ch, PL = (N + 1) << a.log_ch, HN = N >> 1;
    const Tile29 p(lds29, PL);                                            // the tile, transposed: (N+1)*CH elements
    const Tile29 U(lds29 + Tile29::words(PL), HN > 0 ? HN : 1);
    const bool multi = a.n_pass > 1;
    const uint32_t l1 = a.l[0];
    uint64_t r = 0, c0 = 0, Krest = 0;
    uint32_t log_S1 = 0;
    if (multi) {
        const uint64_t tiles_per_r = (1ull << l1) >> a.log_ch;
        r = blockIdx.x / tiles_per_r;
        c0 = (blockIdx.x % tiles_per_r) << a.log_ch;
        Krest = ntt_digit_reverse(a, r, 1, a.n_pass - 2);
        log_S1 = a.log_n - l1;
    }
    for (uint32_t k = threadIdx.x; k < HN; k += NTT29_THREADS) lds29_put<C>(U, k, load29_packed<C>(a.LT + (size_t)k * C::N));
    Fp29<C> sc;
    if (a.scale) sc = load29_packed<C>(a.scale);
    for (uint32_t idx = threadIdx.x; idx < E; idx += NTT29_THREADS) {
        const uint32_t j = idx & (N - 1), c = idx >> l;
        const uint64_t addr = multi ? (((c0 + c) << log_S1) + (r << l) + j) : j;
        Fp29<C> x = IN_REC ? rec29_load<C>(in, addr) : (a.in_len && addr >= a.in_len) ? zero29<C>() : load29_packed<C>(in + addr * C::N);
        if (multi) {
            const uint64_t K = (c0 + c) + (Krest << l1);
            x = mul29(x, ntt29_pow<C>(a, (uint64_t)j * K));
        }
        if (a.rowinc) x = mul29(x, load29_packed<C>(a.rowinc + (size_t)j * C::N));
        if (a.scale) x = mul29(x, sc);
        lds29_put<C>(p, c * (N + 1) + j, x);
    }
    __syncthreads();
    ntt29_stages_t<C>(p, U, l, a.log_ch);
    for (uint32_t idx = threadIdx.x; idx < E; idx += NTT29_THREADS) {
        const uint32_t c = idx & (CH - 1), k = idx >> a.log_ch;
        const uint32_t j = l ? (__brev(k) >> (32 - l)) : 0u;
        const uint64_t K = (c0 + c) + (Krest << l1);
        const uint64_t addr = multi ? (K + ((uint64_t)k << (a.log_n - l))) : k;
        Fp29<C> v = lds29_get<C>(p, c * (N + 1) + j);
        reduce29_small(v);
        uint32_t w[C::N];
        pack29<C>(w, v);
        uint4* q = reinterpret_cast<uint4*>(out + addr * C::N);
        q[0] = make_uint4(w[0], w[1], w[2], w[3]); q[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

}  // namespace zkmi
