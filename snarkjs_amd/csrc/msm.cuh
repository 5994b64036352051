// snarkjs_amd/csrc/msm.cuh — Pippenger multi-scalar multiplication for gfx950.
//
// Replaces ffjavascript's engine_multiexp (window tasks over Web Workers, build/snarkjs.min.js:1@213360/@214651)
// and wasmcurves' g1m_/g2m_multiexpAffine_chunk (@75966): instead of one 2^c-bucket pass per (chunk, window) task
// with the bases re-copied per window, the whole MSM is a few data-parallel device stages over all windows at once:
//
//   1. digit sort     signed-digit recoding of every scalar and, per (window, bucket), the list of (point index, sign):
//                     k_rsort_* (two-level LDS radix partition, large inputs) or k_msm_count/_scan/_scatter (counting sort with
//                     wave-aggregated global atomics, small inputs)
//   2. schedule       k_msm_classify/_assign: load-balanced lane groups (big buckets get 2^j lanes)
//   3. k_msm_accum    gather bases, XYZZ mixed additions (the hot loop: one per non-zero digit), then k_msm_tree/_giant for the
//                     lane partials of multi-lane buckets; with resident bases the gather reads pre-computed window tables
//                     T[k][i] = 2^(c k) P_i (k_msm_precompute) and all digits share ONE set of 2^(c-1) buckets
//   4. reduction      sum_b (b+1) B_b in 2-D form: k_msm_rowcol + k_msm_fold (row / column sums), then k_msm_wsum or k_msm_bitsums
//
// The per-window (or per-bit) sums go back to the host, which folds them with a few doublings (the reference also recombines
// windows on the host, @213360).  Signed digits halve the bucket count: 2^(c-1) buckets per window.
#pragma once
#include "curve.cuh"
#include "field29.cuh"

namespace zkmi {

struct MsmShape {
    uint32_t n;        // terms
    int c;             // window bits
    int Wd;            // digit windows = ceil((8*scalar_bytes + 1) / c)
    int W;             // bucket sets: Wd, or 1 with pre-computed window tables (all digits share one set of buckets)
    uint32_t nb;       // buckets per set = 2^(c-1)
    int sb;            // scalar bytes
    int precomp;       // bases are a table T[k][i] = 2^(c*k) * P_i (k < Wd): entry index = k*stride + i
    uint32_t stride;   // points per table row (>= n: an MSM may use a prefix of the resident bases)
};

// ---- scalar access / signed-digit recoding ----------------------------------------------------------------------
template <int NW> ZK_DEV void load_scalar(uint32_t (&s)[NW], const uint8_t* scalars, size_t i, int sb) {
    const uint8_t* p = scalars + i * (size_t)sb;
    if ((sb & 3) == 0 && ((uintptr_t)scalars & 3) == 0) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
        for (int k = 0; k < NW; k++) s[k] = (4 * k < sb) ? q[k] : 0u;
    } else {
#pragma unroll
        for (int k = 0; k < NW; k++) {
            uint32_t v = 0;
            for (int b = 0; b < 4; b++) if (4 * k + b < sb) v |= (uint32_t)p[4 * k + b] << (8 * b);
            s[k] = v;
        }
    }
}
// bits [bit, bit+c) of the little-endian integer s (zero beyond 32*NW)
template <int NW> ZK_DEV uint32_t window_bits(const uint32_t (&s)[NW], int bit, int c) {
    int wi = bit >> 5, sh = bit & 31;
    uint64_t lo = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        if (k == wi) lo |= s[k];
        if (k == wi + 1) lo |= (uint64_t)s[k] << 32;
    }
    return (uint32_t)(lo >> sh) & ((1u << c) - 1u);
}
// Calls f(window, magnitude in [1, 2^(c-1)], negative) for every non-zero signed digit of s.
template <int NW, class Fn> ZK_DEV void for_each_digit(const uint32_t (&s)[NW], int c, int W, Fn f) {
    uint32_t carry = 0;
    const uint32_t half = 1u << (c - 1);
    for (int w = 0; w < W; w++) {
        uint32_t raw = window_bits<NW>(s, w * c, c) + carry;
        bool neg = raw > half;
        uint32_t mag = neg ? ((1u << c) - raw) : raw;
        carry = neg ? 1u : 0u;
        if (mag) f(w, mag, neg);
    }
}

// atomicAdd(&ctr[idx], 1) for every active lane, returning the old value, with up to two rounds of wave-level
// aggregation: lanes that share the first pending lane's counter are served by one atomic. Skewed digit distributions
// (real witnesses are mostly 0/1; the partially filled top window) otherwise serialise on a single L2 address.
ZK_DEV uint32_t msm_atomic_inc(uint32_t* __restrict__ ctr, size_t idx) {
    const uint32_t lane = __lane_id();
    uint32_t result = 0;
    bool done = false;
    // Each round serves the group of lanes that share the first pending lane's counter with ONE atomic. Rounds continue
    // while groups are large (a hot digit: up to 8 distinct hot values per wave), and stop at the first small group —
    // uniformly distributed digits leave after one round.
    for (int round = 0; round < 8; round++) {
        const uint64_t pend = __ballot(!done);
        if (!pend) break;
        const int leader = __ffsll((unsigned long long)pend) - 1;
        const uint32_t lo = __shfl((uint32_t)idx, leader), hi = __shfl((uint32_t)(idx >> 32), leader);
        const bool match = !done && lo == (uint32_t)idx && hi == (uint32_t)(idx >> 32);
        const uint64_t peers = __ballot(match);
        uint32_t basev = 0;
        if (match && (int)lane == leader) basev = atomicAdd(&ctr[idx], (uint32_t)__popcll(peers));
        basev = __shfl(basev, leader);
        if (match) { result = basev + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull)); done = true; }
        if (__popcll(peers) < 4) break;
    }
    if (!done) result = atomicAdd(&ctr[idx], 1u);
    return result;
}

// dropmask (optional): bit i set = scalar i is dropped (its base is the point at infinity in every MSM run over this plan)
template <int NW> __global__ void k_msm_count(const uint8_t* __restrict__ scalars, MsmShape sh, const uint32_t* __restrict__ dropmask, uint32_t* __restrict__ counts) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sh.n) return;
    if (dropmask && ((dropmask[i >> 5] >> (i & 31)) & 1u)) return;
    uint32_t s[NW];
    load_scalar<NW>(s, scalars, i, sh.sb);
    for_each_digit<NW>(s, sh.c, sh.Wd, [&](int w, uint32_t mag, bool) { msm_atomic_inc(counts, (sh.precomp ? (size_t)0 : (size_t)w * sh.nb) + (mag - 1)); });
}

// starts[] = exclusive prefix sum of counts[] over ALL (window, bucket) pairs: the sorted lists of all buckets form one flat
// array. Three small launches: per-chunk sums, scan of the chunk sums (one block), per-chunk scan with offset.
constexpr uint32_t MSM_SCAN_CHUNK = 2048;      // 256 threads x 8 counters
static __global__ void __launch_bounds__(256) k_msm_scan_sums(const uint32_t* __restrict__ counts, uint32_t total, uint32_t* __restrict__ part) {
    __shared__ uint32_t red[256];
    const uint32_t base = blockIdx.x * MSM_SCAN_CHUNK + threadIdx.x * 8;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) if (base + k < total) s += counts[base + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) { if (threadIdx.x < (uint32_t)d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
static __global__ void __launch_bounds__(1024) k_msm_scan_top(uint32_t* __restrict__ part, uint32_t nparts) {
    __shared__ uint32_t sh[1024];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nparts; b0 += 1024) {
        const uint32_t i = b0 + threadIdx.x, v = i < nparts ? part[i] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            uint32_t t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nparts) part[i] = carry + sh[threadIdx.x] - v;
        const uint32_t tot = sh[1023];
        __syncthreads();
        carry += tot;
    }
}
static __global__ void __launch_bounds__(256) k_msm_scan_final(const uint32_t* __restrict__ counts, uint32_t total, const uint32_t* __restrict__ part, uint32_t* __restrict__ starts) {
    __shared__ uint32_t sh[256];
    const uint32_t base = blockIdx.x * MSM_SCAN_CHUNK + threadIdx.x * 8;
    uint32_t c[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { c[k] = base + k < total ? counts[base + k] : 0u; s += c[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = part[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 8; k++) { if (base + k < total) starts[base + k] = run; run += c[k]; }
}

template <int NW> __global__ void k_msm_scatter(const uint8_t* __restrict__ scalars, MsmShape sh, const uint32_t* __restrict__ dropmask, const uint32_t* __restrict__ starts,
                                               uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sh.n) return;
    if (dropmask && ((dropmask[i >> 5] >> (i & 31)) & 1u)) return;
    uint32_t s[NW];
    load_scalar<NW>(s, scalars, i, sh.sb);
    for_each_digit<NW>(s, sh.c, sh.Wd, [&](int w, uint32_t mag, bool neg) {
        const size_t g = (sh.precomp ? (size_t)0 : (size_t)w * sh.nb) + (mag - 1);
        const uint32_t pos = starts[g] + msm_atomic_inc(cursor, g);
        const uint32_t ent = sh.precomp ? (uint32_t)w * sh.stride + (uint32_t)i : (uint32_t)i;
        sorted[pos] = ent | (neg ? 0x80000000u : 0u);
    });
}

// ---- two-level LDS radix partition (large MSMs) -----------------------------------------------------------------------
// The counting sort above issues one GLOBAL atomic per (scalar, digit) twice (count, scatter): 2 x 13.6M atomics per 2^20-term MSM
// with window tables, ~1.3 ms. For large inputs the same lists are built with LDS atomics only, in two passes over the flat key
// g = window*nb + (mag-1) (window = 0 with tables):
//   level 1  partition by g >> 11: per-block LDS histograms (k_rsort_hist1), one flat scan, block-local ranking (k_rsort_scatter1)
//            into an intermediate array of (entry, g & 2047) pairs;
//   level 2  every partition is cut into chunks of <= RSORT_CHUNK pairs (k_rsort_chunks); per-chunk LDS histogram over the 2048
//            low keys (k_rsort_hist2), per-partition scan over chunks and bins (k_rsort_scan2: this also produces counts[] and
//            starts[]), per-chunk ranking into the final lists (k_rsort_scatter2).
// Any distribution works: a partition that receives most of the entries (real witnesses: digit 1 of window 0) simply gets many
// chunks. The order inside a bucket is arbitrary (as with the atomic version); bucket sums do not depend on it.
constexpr int RSORT_LOW_BITS = 11;
constexpr uint32_t RSORT_BINS = 1u << RSORT_LOW_BITS;
constexpr uint32_t RSORT_TILE = 1024;            // scalars per level-1 block (256 threads x 4)
constexpr uint32_t RSORT_CHUNK = 8192;           // pairs per level-2 block
constexpr uint32_t RSORT_MAX_PARTS = 4096;

template <int NW, int BS = 256, class Fn> ZK_DEV void rsort_tile_digits(const uint8_t* __restrict__ scalars, const MsmShape& sh, const uint32_t* __restrict__ dropmask, Fn f) {
#pragma unroll 1
    for (uint32_t j = 0; j < RSORT_TILE / BS; j++) {
        const size_t i = (size_t)blockIdx.x * RSORT_TILE + j * BS + threadIdx.x;
        if (i >= sh.n) break;
        if (dropmask && ((dropmask[i >> 5] >> (i & 31)) & 1u)) continue;
        uint32_t s[NW];
        load_scalar<NW>(s, scalars, i, sh.sb);
        for_each_digit<NW>(s, sh.c, sh.Wd, [&](int w, uint32_t mag, bool neg) {
            const uint32_t g = (sh.precomp ? 0u : (uint32_t)w * sh.nb) + (mag - 1);
            const uint32_t ent = (sh.precomp ? (uint32_t)w * sh.stride + (uint32_t)i : (uint32_t)i) | (neg ? 0x80000000u : 0u);
            f(g, ent);
        });
    }
}
// bh[p * nblk + blk] = entries of block blk that fall into partition p
template <int NW> __global__ void __launch_bounds__(256)
k_rsort_hist1(const uint8_t* __restrict__ scalars, MsmShape sh, const uint32_t* __restrict__ dropmask, uint32_t nparts, uint32_t lb, uint32_t* __restrict__ bh) {
    __shared__ uint32_t h[RSORT_MAX_PARTS];
    for (uint32_t p = threadIdx.x; p < nparts; p += 256) h[p] = 0;
    __syncthreads();
    rsort_tile_digits<NW>(scalars, sh, dropmask, [&](uint32_t g, uint32_t) { atomicAdd(&h[g >> lb], 1u); });
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < nparts; p += 256) bh[(size_t)p * gridDim.x + blockIdx.x] = h[p];
    if (blockIdx.x == 0 && threadIdx.x == 0) bh[(size_t)nparts * gridDim.x] = 0;              // the scan's closing entry (r06: was a fill launch of its own)
}
template <int NW> __global__ void __launch_bounds__(256)
k_rsort_scatter1(const uint8_t* __restrict__ scalars, MsmShape sh, const uint32_t* __restrict__ dropmask, uint32_t nparts, uint32_t lb, const uint32_t* __restrict__ bhoff, uint2* __restrict__ tmp) {
    __shared__ uint32_t cur[RSORT_MAX_PARTS];
    for (uint32_t p = threadIdx.x; p < nparts; p += 256) cur[p] = bhoff[(size_t)p * gridDim.x + blockIdx.x];
    __syncthreads();
    rsort_tile_digits<NW>(scalars, sh, dropmask, [&](uint32_t g, uint32_t ent) {
        const uint32_t pos = atomicAdd(&cur[g >> lb], 1u);
        tmp[pos] = make_uint2(ent, g & ((1u << lb) - 1));
    });
}
// r06: the same scatter with the block's pairs ranked in LDS first and written out in partition order. The direct version above issues one 8-byte store per
// pair to wherever its partition's cursor stands: 64 lanes, 64 partitions, 64 write transactions per wave instruction — 13.6 M of them for a 2^20-term table
// MSM, 109 us standalone against 35 us for k_rsort_hist1, which does the same digit work. Here a lane of the copy-out loop writes the pair next to its
// neighbour's: a block's share of a partition (26 pairs on uniform scalars) leaves as one contiguous run. This block's count per partition is the difference
// of two neighbouring entries of the scanned matrix (entry p nblk + blk + 1 follows entry p nblk + blk in scan order, across the partition boundary too).
// LDS: cursor[P] | delta[P] | scan[1024] | entry[cap] | key[cap], cap = Wd x RSORT_TILE pairs (110 KB for 13 digits and 512 partitions: one 1024-lane block per CU).
template <int NW> __global__ void __launch_bounds__(1024)
k_rsort_scatter1_staged(const uint8_t* __restrict__ scalars, MsmShape sh, const uint32_t* __restrict__ dropmask, uint32_t nparts, uint32_t lb, const uint32_t* __restrict__ bhoff,
                        uint32_t cap, uint2* __restrict__ tmp) {
    extern __shared__ uint32_t rs1_lds[];
    uint32_t *cur = rs1_lds, *delta = cur + nparts, *sc = delta + nparts, *s_ent = sc + 1024, *s_key = s_ent + cap;
    // partitions per lane of the block scan (nparts <= RSORT_MAX_PARTS = 4 x 1024)
    const uint32_t per = (nparts + 1023) / 1024, p0 = threadIdx.x * per;
    uint32_t gb[4], cn[4], sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        gb[q] = 0; cn[q] = 0;
        if (q < per && p0 + q < nparts) {
            const size_t at = (size_t)(p0 + q) * gridDim.x + blockIdx.x;
            gb[q] = bhoff[at]; cn[q] = bhoff[at + 1] - gb[q];
            sum += cn[q];
        }
    }
    sc[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t t = threadIdx.x >= d ? sc[threadIdx.x - d] : 0u;
        __syncthreads();
        sc[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = sc[threadIdx.x] - sum;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) if (q < per && p0 + q < nparts) { cur[p0 + q] = run; delta[p0 + q] = gb[q] - run; run += cn[q]; }
    const uint32_t mine = sc[1023];
    __syncthreads();
    rsort_tile_digits<NW, 1024>(scalars, sh, dropmask, [&](uint32_t g, uint32_t ent) {
        const uint32_t pos = atomicAdd(&cur[g >> lb], 1u);
        s_ent[pos] = ent; s_key[pos] = g;
    });
    __syncthreads();
    const uint32_t low = (1u << lb) - 1;
    for (uint32_t j = threadIdx.x; j < mine; j += 1024) {
        const uint32_t g = s_key[j];
        tmp[delta[g >> lb] + j] = make_uint2(s_ent[j], g & low);
    }
}
// chunk table: chunks[3k..3k+2] = (partition, first pair, number of pairs); meta[0] = number of chunks. One block.
static __global__ void __launch_bounds__(1024)
k_rsort_chunks(const uint32_t* __restrict__ bhoff, uint32_t nparts, uint32_t nblk, uint32_t fused_cap, uint32_t* __restrict__ pchunk0, uint32_t* __restrict__ chunks, uint32_t* __restrict__ meta,
               uint32_t* __restrict__ zero, uint32_t zero_words) {
    __shared__ uint32_t sc[1024];
    for (uint32_t i = threadIdx.x; i < zero_words; i += 1024) zero[i] = 0;                       // the lane scheduler's histogram / cursors / meta words (r06: was a fill launch of its own)
    uint32_t carry = 0;
    for (uint32_t p0 = 0; p0 < nparts; p0 += 1024) {
        const uint32_t p = p0 + threadIdx.x;
        uint32_t start = 0, size = 0;
        if (p < nparts) { start = bhoff[(size_t)p * nblk]; size = bhoff[(size_t)(p + 1) * nblk] - start; }
        // fused_cap != 0: partitions of at most that many pairs are finished by k_rsort_part alone and get no chunks
        const uint32_t nch = (fused_cap && size <= fused_cap) ? 0u : (size + RSORT_CHUNK - 1) / RSORT_CHUNK;
        sc[threadIdx.x] = nch;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            uint32_t t = threadIdx.x >= d ? sc[threadIdx.x - d] : 0u;
            __syncthreads();
            sc[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t c0 = carry + sc[threadIdx.x] - nch;
        if (p < nparts) {
            pchunk0[p] = c0;
            for (uint32_t k = 0; k < nch; k++) {
                chunks[3 * (c0 + k)] = p; chunks[3 * (c0 + k) + 1] = start + k * RSORT_CHUNK;
                chunks[3 * (c0 + k) + 2] = min(RSORT_CHUNK, size - k * RSORT_CHUNK);
            }
        }
        const uint32_t tot = sc[1023];
        __syncthreads();
        carry += tot;
    }
    if (threadIdx.x == 0) { pchunk0[nparts] = carry; meta[0] = carry; }
}
static __global__ void __launch_bounds__(256)
k_rsort_hist2(const uint2* __restrict__ tmp, const uint32_t* __restrict__ chunks, const uint32_t* __restrict__ meta, uint32_t* __restrict__ h2) {
    // r06: the grid is a few hundred blocks walking the chunk table, not one block per POSSIBLE chunk: with the fused level 2 (k_rsort_part) the table is
    // usually empty, and 2 177 blocks that read meta[0] and leave cost 17 us per launch
    __shared__ uint32_t h[RSORT_BINS];
    const uint32_t nch = meta[0];
    for (uint32_t ch = blockIdx.x; ch < nch; ch += gridDim.x) {
        for (uint32_t b = threadIdx.x; b < RSORT_BINS; b += 256) h[b] = 0;
        __syncthreads();
        const uint32_t first = chunks[3 * ch + 1], len = chunks[3 * ch + 2];
        for (uint32_t j = threadIdx.x; j < len; j += 256) atomicAdd(&h[tmp[first + j].y], 1u);
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < RSORT_BINS; b += 256) h2[(size_t)ch * RSORT_BINS + b] = h[b];
        __syncthreads();
    }
}
// one block per partition: h2[chunk][bin] <- exclusive prefix over the partition's chunks; counts / starts of the partition's buckets
static __global__ void __launch_bounds__(1024)
k_rsort_scan2(const uint32_t* __restrict__ bhoff, uint32_t nblk, uint32_t lb, uint32_t fused, const uint32_t* __restrict__ pchunk0, uint32_t* __restrict__ h2, uint32_t* __restrict__ counts, uint32_t* __restrict__ starts) {
    __shared__ uint32_t sc[1024];
    const uint32_t p = blockIdx.x, c0 = pchunk0[p], c1 = pchunk0[p + 1], bins = 1u << lb;
    if (fused && c0 == c1) return;                       // no chunks: k_rsort_part wrote this partition's counts / starts (an empty partition included)
    const uint32_t b0 = 2 * threadIdx.x;                 // two adjacent bins per thread
    uint32_t r0 = 0, r1 = 0;
    for (uint32_t k = c0; k < c1; k++) {
        uint2* q = reinterpret_cast<uint2*>(h2 + (size_t)k * RSORT_BINS + b0);
        const uint2 t = *q;
        *q = make_uint2(r0, r1);
        r0 += t.x; r1 += t.y;
    }
    sc[threadIdx.x] = r0 + r1;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t t = threadIdx.x >= d ? sc[threadIdx.x - d] : 0u;
        __syncthreads();
        sc[threadIdx.x] += t;
        __syncthreads();
    }
    const uint32_t base = bhoff[(size_t)p * nblk] + sc[threadIdx.x] - (r0 + r1);
    if (b0 < bins) {                                     // bins beyond 2^lb hold nothing (low keys are < 2^lb)
        const size_t g = (size_t)p * bins + b0;
        counts[g] = r0; counts[g + 1] = r1;
        starts[g] = base; starts[g + 1] = base + r0;
    }
}
static __global__ void __launch_bounds__(256)
k_rsort_scatter2(const uint2* __restrict__ tmp, const uint32_t* __restrict__ chunks, const uint32_t* __restrict__ meta, const uint32_t* __restrict__ h2,
                 const uint32_t* __restrict__ starts, uint32_t lb, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t cur[RSORT_BINS];
    const uint32_t nch = meta[0], bins = 1u << lb;
    for (uint32_t ch = blockIdx.x; ch < nch; ch += gridDim.x) {                 // r06: a walk over the chunk table, as in k_rsort_hist2
        const uint32_t p = chunks[3 * ch], first = chunks[3 * ch + 1], len = chunks[3 * ch + 2];
        for (uint32_t b = threadIdx.x; b < bins; b += 256) cur[b] = starts[(size_t)p * bins + b] + h2[(size_t)ch * RSORT_BINS + b];
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < len; j += 256) {
            const uint2 e = tmp[first + j];
            sorted[atomicAdd(&cur[e.y], 1u)] = e.x;
        }
        __syncthreads();
    }
}
// r05: level 2 in ONE kernel for a partition whose pairs fit an LDS staging buffer (one block per partition): histogram of the low keys, scan
// (this writes the partition's counts[] / starts[]), ranking into LDS, and a COALESCED copy of the finished lists — instead of per-chunk
// histograms in global memory, a per-partition scan kernel and a scatter whose 4-byte stores land 2^lb lists apart (k_rsort_hist2 / _scan2 /
// _scatter2: 0.28 ms of the 0.75 ms a 13.6 M-entry sort takes inside a PLONK proof). Partitions beyond `cap` pairs (skewed scalar
// distributions, MSMs beyond ~2^20 terms) return at once and are done by the chunked kernels as before (k_rsort_chunks gives only them chunks).
// LDS: [2^lb] histogram, then cursors | [1024] scan | [cap] staged entries.
static __global__ void __launch_bounds__(1024)
k_rsort_part(const uint2* __restrict__ tmp, const uint32_t* __restrict__ bhoff, uint32_t nblk, uint32_t lb, uint32_t cap, uint32_t* __restrict__ counts, uint32_t* __restrict__ starts,
             uint32_t* __restrict__ sorted) {
    extern __shared__ uint32_t rs_lds[];
    const uint32_t p = blockIdx.x, bins = 1u << lb;
    const uint32_t first = bhoff[(size_t)p * nblk], size = bhoff[(size_t)(p + 1) * nblk] - first;
    if (size > cap) return;
    uint32_t *h = rs_lds, *sc = rs_lds + bins, *stage = sc + 1024;
    for (uint32_t b = threadIdx.x; b < bins; b += 1024) h[b] = 0;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < size; j += 1024) atomicAdd(&h[tmp[first + j].y], 1u);
    __syncthreads();
    const uint32_t b0 = 2 * threadIdx.x;                 // two adjacent bins per thread (bins <= 2048)
    const uint32_t r0 = b0 < bins ? h[b0] : 0u, r1 = b0 + 1 < bins ? h[b0 + 1] : 0u;
    sc[threadIdx.x] = r0 + r1;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t t = threadIdx.x >= d ? sc[threadIdx.x - d] : 0u;
        __syncthreads();
        sc[threadIdx.x] += t;
        __syncthreads();
    }
    const uint32_t excl = sc[threadIdx.x] - (r0 + r1);
    if (b0 < bins) {
        const size_t g = (size_t)p * bins + b0;
        counts[g] = r0; counts[g + 1] = r1;
        starts[g] = first + excl; starts[g + 1] = first + excl + r0;
        h[b0] = excl; h[b0 + 1] = excl + r0;             // cursors, relative to the partition's first entry
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < size; j += 1024) {
        const uint2 e = tmp[first + j];
        stage[atomicAdd(&h[e.y], 1u)] = e.x;
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < size; j += 1024) sorted[first + j] = stage[j];
}

// ---- bucket accumulation: load-balanced lane groups ---------------------------------------------------------------
// A bucket with cnt points gets L lanes: L = 1 while cnt < 2*cap, else L = 2^j with j = floor(log2(cnt/cap)) (so every
// lane adds < 2*cap points).  Buckets are ordered by a key (descending): multi-lane groups first (largest first, so a
// group of 2^j lanes always starts at a multiple of 2^j), then single-lane buckets by exact size so that the lanes of a
// wave run equally long loops.  Scalar distributions with huge buckets (real witnesses are mostly 0/1; the partial top
// window) therefore cost the same as uniform ones.  Group partial sums are combined by k_msm_tree / k_msm_giant.
constexpr uint32_t MSM_MAX_CAP = 256;
constexpr uint32_t MSM_NKEYS = 2 * MSM_MAX_CAP + 32;
// meta words: [0] total lanes, [1] lanes that belong to multi-lane groups, [2] number of giant buckets (> one tree block)
ZK_DEV uint32_t msm_key(uint32_t cnt, uint32_t cap) {
    if (cnt < 2 * cap) return cnt;                                   // 0 = empty, else single lane, key = size
    return 2 * cap - 1 + (31 - __clz(cnt / cap));                    // j >= 1
}
ZK_DEV uint32_t msm_key_lanes_log(uint32_t key, uint32_t cap) { return key < 2 * cap ? 0u : key - (2 * cap - 1); }

// every coordinate times 2^5 (2^8 for BLS12-381): the same point with its coordinates moved from the reference's R-form (x 2^(32 N)) to the
// R'-form of field29.cuh (x 2^(B NL)) — buckets whose row / column sums are formed on unsaturated limbs (msm29.cuh: k_msm_rowcol_wave29) are
// kept in that form
template <class F> ZK_DEV void pt_scale32(XYZZ<F>& p) {
#pragma unroll 1
    for (int k = 0; k < r29_shift<typename F::Cfg>(); k++) { p.X = f_dbl(p.X); p.Y = f_dbl(p.Y); p.ZZ = f_dbl(p.ZZ); p.ZZZ = f_dbl(p.ZZZ); }
}
// r05: every block walks a contiguous run of buckets (grid-stride inside the run) instead of 256 buckets per block: with uniform scalars the bucket sizes
// fall into ~30 key classes, and 2 048 blocks each adding its ~30 class counts to the same ~30 global words serialised in the L2 (classify 57 us,
// assign 115 us per sort of a 2^20-term MSM, r04 trace); a few hundred blocks issue a sixteenth of those atomics.
constexpr uint32_t MSM_SCHED_BLOCKS = 256;
ZK_DEV void msm_sched_run(uint32_t total, uint32_t& lo, uint32_t& hi) {
    const uint32_t per = ((total + gridDim.x - 1) / gridDim.x + 255u) & ~255u;
    lo = min(total, blockIdx.x * per);
    hi = min(total, lo + per);
}
static __global__ void __launch_bounds__(256) k_msm_classify(const uint32_t* __restrict__ counts, uint32_t total, uint32_t cap, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[MSM_NKEYS];
    for (uint32_t i = threadIdx.x; i < MSM_NKEYS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    uint32_t lo, hi;
    msm_sched_run(total, lo, hi);
    for (uint32_t g = lo + threadIdx.x; g < hi; g += blockDim.x) { const uint32_t k = msm_key(counts[g], cap); if (k) atomicAdd(&h[k], 1u); }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < MSM_NKEYS; i += blockDim.x) if (h[i]) atomicAdd(&hist[i], h[i]);
}
// r06: the first lane of every key class (k_msm_class_scan's table) is recomputed by every block in its own LDS by its first wave, against a launch
// of its own between classify and assign (19 us of stream time per sort, nine sorts per PLONK proof); block 0 publishes the totals
static __global__ void __launch_bounds__(256)
k_msm_assign(const uint32_t* __restrict__ counts, uint32_t total, uint32_t cap, uint32_t log_tb, const uint32_t* __restrict__ hist, uint32_t* __restrict__ cursor,
             uint32_t* __restrict__ lane_g, uint32_t* __restrict__ lane_sub, uint32_t* __restrict__ giants, uint32_t* __restrict__ meta) {
    __shared__ uint32_t h[MSM_NKEYS], base[MSM_NKEYS], off[MSM_NKEYS + 1];
    for (uint32_t i = threadIdx.x; i < MSM_NKEYS; i += blockDim.x) { h[i] = 0; off[i] = i ? hist[i] << msm_key_lanes_log(i, cap) : 0u; }
    __syncthreads();
    if (threadIdx.x < 64) {
        // exclusive suffix scan over the classes in descending key order by ONE wave: lane L owns the KPL keys below MSM_NKEYS - KPL L (the walk of
        // one lane over all 544 classes was 15 us of every block's start)
        constexpr int KPL = (MSM_NKEYS + 63) / 64;
        const int top = (int)MSM_NKEYS - 1 - KPL * (int)threadIdx.x;
        uint32_t s = 0;
#pragma unroll
        for (int q = 0; q < KPL; q++) { const int k = top - q; if (k >= 1) s += off[k]; }
        uint32_t incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if ((int)threadIdx.x >= d) incl += t; }
        uint32_t run = incl - s;
#pragma unroll
        for (int q = 0; q < KPL; q++) { const int k = top - q; if (k >= 1) { const uint32_t c = off[k]; off[k] = run; run += c; } }
        const uint32_t all = __shfl(incl, 63);
        // lanes of multi-lane groups = classes 2 cap and above = the first lane of class 2 cap - 1 (cap >= 8: that class exists and is a single-lane one)
        if (blockIdx.x == 0 && threadIdx.x == 0) { meta[0] = all; meta[1] = off[2 * cap - 1]; }
    }
    uint32_t lo, hi;
    msm_sched_run(total, lo, hi);
    // pass 1: this block's class counts; ONE global atomic per class present reserves the block's ranks; pass 2 hands them out (any bijection
    // will do: the order of the buckets inside a class is not part of the result)
    for (uint32_t g = lo + threadIdx.x; g < hi; g += blockDim.x) { const uint32_t k = msm_key(counts[g], cap); if (k) atomicAdd(&h[k], 1u); }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < MSM_NKEYS; i += blockDim.x) { if (h[i]) base[i] = atomicAdd(&cursor[i], h[i]); h[i] = 0; }
    __syncthreads();
    for (uint32_t g = lo + threadIdx.x; g < hi; g += blockDim.x) {
        const uint32_t k = msm_key(counts[g], cap);
        if (!k) continue;
        const uint32_t j = msm_key_lanes_log(k, cap), rank = base[k] + atomicAdd(&h[k], 1u);
        const uint32_t lane0 = off[k] + (rank << j);
        for (uint32_t l = 0; l < (1u << j); l++) { lane_g[lane0 + l] = g; lane_sub[lane0 + l] = l; }
        if (j > log_tb) {
            uint32_t idx = atomicAdd(&meta[2], 1u);
            giants[3 * idx] = g; giants[3 * idx + 1] = lane0 >> log_tb; giants[3 * idx + 2] = 1u << (j - log_tb);
        }
    }
}
// Threads per accumulation block. The LDS-parked Fq2 accumulators cost 4*FW words per lane: 256 B (BN254: 2 blocks of 256 lanes
// per CU) but 384 B for BLS12-381, where a 256-lane block would own 96 KiB and run alone on its CU (1 wave per SIMD) — 128-lane
// blocks fit three per CU.
template <class F> struct MsmAccumBlock { static constexpr int value = (FieldWords<F>::value > 16) ? 128 : 256; };
// WIDE (Fq2 points): at most 256 VGPRs (2 waves per SIMD), accumulator parked in LDS (curve.cuh: LdsAcc), no software
// pipelining of the gather — 2.6x the throughput of the fully inlined 400+-register version (tools/maddbench.hip).
template <class F, bool WIDE, bool MERGE> __global__ void __launch_bounds__(256, WIDE ? 2 : 1)
k_msm_accum(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ infmask, MsmShape sh, uint32_t skip, uint32_t cap, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ starts,
            const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ lane_g, const uint32_t* __restrict__ lane_sub, const uint32_t* __restrict__ meta,
            uint32_t* __restrict__ buckets, uint32_t* __restrict__ lane_partials, const uint32_t* __restrict__ prev_counts) {
    constexpr int FW = FieldWords<F>::value;
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= meta[0]) return;
    const uint32_t g = lane_g[lane];
    const uint32_t w = g / sh.nb;
    const uint32_t cnt = counts[g];
    const uint32_t j = msm_key_lanes_log(msm_key(cnt, cap), cap);
    uint32_t lo = 0, hi = cnt;
    if (j) {
        const uint32_t chunk = (cnt + (1u << j) - 1) >> j;
        lo = min(cnt, lane_sub[lane] * chunk);
        hi = min(cnt, lo + chunk);
    }
    const uint32_t* list = sorted + starts[g];
    XYZZ<F> acc;
    pt_set_inf(acc);
    // merge mode (prev_counts != nullptr): the bucket array already holds the buckets of another MSM of the same shape whose
    // result is only ever needed added to this one (Groth16: C and H) — the first lane of each bucket continues from that value,
    // so the two MSMs share ONE bucket reduction
    if (MERGE && !WIDE) { if (prev_counts[g] && (!j || lane_sub[lane] == 0)) pt_load(acc, buckets + (size_t)g * (4 * FW)); }
    // Next usable entry of this lane's list: skips indices below `skip` and bases at infinity (zkey sections B1/B2 are mostly
    // infinity for real circuits) in a cheap private loop, so that every lane arrives at the mixed addition with a real point —
    // a lane that merely `continue`d would idle for the whole addition of its 63 neighbours. infmask: one bit per table entry
    // (resident tables), else the point itself is inspected.
    uint32_t k = lo;
    auto fetch = [&](uint32_t& e_out, Affine<F>& q_out) -> bool {
        while (k < hi) {
            const uint32_t e = list[k++];
            uint32_t idx = e & 0x7fffffffu;
            if (idx < skip) continue;
            idx -= skip;
            if (infmask) {
                if ((infmask[idx >> 5] >> (idx & 31)) & 1u) continue;
                pt_load(q_out, bases + (size_t)idx * (2 * FW));
            } else {
                pt_load(q_out, bases + (size_t)idx * (2 * FW));
                if (pt_is_inf(q_out)) continue;
            }
            e_out = e;
            return true;
        }
        return false;
    };
    if (WIDE) {
        extern __shared__ __attribute__((aligned(16))) uint32_t lds_acc[];
        LdsAcc<F, MsmAccumBlock<F>::value> A{lds_acc + threadIdx.x};
        bool inf = true;
        uint32_t e;
        Affine<F> q;
        while (fetch(e, q)) {
            if (e >> 31) q.y = f_neg(q.y);
            pt_madd_lds(A, inf, q);
        }
        if (!inf) { A.get(0, acc.X); A.get(1, acc.Y); A.get(2, acc.ZZ); A.get(3, acc.ZZZ); }
    } else {
        // software pipeline: the gather of the next point (a random 64..96-byte read) is in flight during the current addition
        uint32_t e_next = 0;
        Affine<F> q_next;
        bool have = fetch(e_next, q_next);
        while (have) {
            const uint32_t e = e_next;
            Affine<F> q = q_next;
            have = fetch(e_next, q_next);
            if (e >> 31) q.y = f_neg(q.y);
            pt_madd(acc, q);
        }
    }
    if (j) pt_store(lane_partials + (size_t)lane * (4 * FW), acc);
    else pt_store(buckets + (size_t)g * (4 * FW), acc);
}
// Diagnostics (zkmi_msm_stats): the number of mixed additions an accumulation launch performs = list entries whose base index is >= skip and
// not flagged in the infinity bitmap. bench.py's int_alu figures use this count instead of a model of the scalar distribution.
static __global__ void __launch_bounds__(256)
k_msm_count_adds(const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ starts, uint32_t total, uint32_t skip,
                 const uint32_t* __restrict__ infmask, unsigned long long* __restrict__ out) {
    const uint32_t E = starts[total - 1] + counts[total - 1];
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x) {
        uint32_t idx = sorted[i] & 0x7fffffffu;
        if (idx < skip) continue;
        idx -= skip;
        if (infmask && ((infmask[idx >> 5] >> (idx & 31)) & 1u)) continue;
        c++;
    }
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}
static __global__ void k_msm_counts_add(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t total, uint32_t* __restrict__ out) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < total) out[g] = (a[g] || b[g]) ? 1u : 0u;
}
// Combine the lane partials of multi-lane groups (they occupy lanes [0, meta[1])): LDS tree inside each block of TB
// lanes; groups wider than a block leave one partial per block for k_msm_giant.
template <class F, int TB> __global__ void __launch_bounds__(TB)
k_msm_tree(const uint32_t* __restrict__ lane_partials, const uint32_t* __restrict__ lane_g, const uint32_t* __restrict__ counts, uint32_t cap,
           const uint32_t* __restrict__ meta, uint32_t* __restrict__ buckets, uint32_t* __restrict__ block_partials, int bucket_x32) {
    constexpr int PW = 4 * FieldWords<F>::value;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t jmax_s;
    // persistent blocks: the grid is a fixed few hundred blocks that stride over the tree blocks actually in use (none at all for
    // uniformly distributed scalars) — a grid sized for the worst case costs ~100 us of wave launches with scratch set-up
    const uint32_t t = threadIdx.x, multi = meta[1];
    for (uint32_t blk = blockIdx.x; blk * TB < multi; blk += gridDim.x) {
        const uint32_t lane = blk * TB + t;
        XYZZ<F> acc;
        uint32_t j = 0, g = 0;
        if (lane < multi) {
            g = lane_g[lane];
            j = msm_key_lanes_log(msm_key(counts[g], cap), cap);
            pt_load(acc, lane_partials + (size_t)lane * PW);
        } else pt_set_inf(acc);
        if (t == 0) jmax_s = j;
        pt_store(lds + t * PW, acc);
        __syncthreads();
        constexpr uint32_t LOG_TB = (TB == 256) ? 8 : (TB == 128 ? 7 : 6);
        const uint32_t steps = min(jmax_s, LOG_TB);
        for (uint32_t s = 0; s < steps; s++) {
            const bool act = ((t & ((2u << s) - 1)) == 0) && (s < j);
            if (act) { XYZZ<F> o; pt_load(o, lds + (t + (1u << s)) * PW); acc = pt_add(acc, o); }
            __syncthreads();
            if (act) pt_store(lds + t * PW, acc);
            __syncthreads();
        }
        if (lane < multi) {
            if (j <= LOG_TB) { if ((t & ((1u << j) - 1)) == 0) { if (bucket_x32) pt_scale32(acc); pt_store(buckets + (size_t)g * PW, acc); } }
            else if (t == 0) pt_store(block_partials + (size_t)blk * PW, acc);
        }
        __syncthreads();
    }
}
template <class F, int TB> __global__ void __launch_bounds__(TB)
k_msm_giant(const uint32_t* __restrict__ giants, const uint32_t* __restrict__ meta, const uint32_t* __restrict__ block_partials, uint32_t* __restrict__ buckets, int bucket_x32) {
    constexpr int PW = 4 * FieldWords<F>::value;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t t = threadIdx.x, ngiants = meta[2];
    for (uint32_t gi = blockIdx.x; gi < ngiants; gi += gridDim.x) {
        const uint32_t g = giants[3 * gi], b0 = giants[3 * gi + 1], nblk = giants[3 * gi + 2];
        XYZZ<F> acc;
        pt_set_inf(acc);
        for (uint32_t b = t; b < nblk; b += TB) { XYZZ<F> o; pt_load(o, block_partials + (size_t)(b0 + b) * PW); acc = pt_add(acc, o); }
        pt_store(lds + t * PW, acc);
        __syncthreads();
        for (int d = TB / 2; d >= 1; d >>= 1) {
            if (t < (uint32_t)d) { XYZZ<F> o; pt_load(o, lds + (t + d) * PW); acc = pt_add(acc, o); pt_store(lds + t * PW, acc); }
            __syncthreads();
        }
        if (t == 0) { if (bucket_x32) pt_scale32(acc); pt_store(buckets + (size_t)g * PW, acc); }
        __syncthreads();
    }
}

// ---- bucket reduction -------------------------------------------------------------------------------------------------
// Per window S = sum_{b0=0}^{nb-1} (b0+1) * B_b0 (b0+1 = digit magnitude). The reference does this with a recursive halving
// (_reduceTable, min.js:1@74634); a GPU wants the shallowest dependency chain, because one XYZZ addition is ~9 us of
// latency for a lone wave. Write b0 = r*C + c with C = 2^cbits columns and R = 2^rbits rows (rbits + cbits = c-1):
//     S = C * sum_r r*Row_r + sum_c c*Col_c + sum_r Row_r,     Row_r = sum_c B[r*C+c],  Col_c = sum_r B[r*C+c].
// k_msm_rowcol forms all Row/Col sums (L lanes per sum: sequential partial sums, then a log2(L)-level butterfly);
// k_msm_wsum turns each length-C array into (sum_t t*X_t, sum_t X_t) with a suffix scan + tree in LDS. The host applies
// the factor C (cbits doublings) when it folds the windows.  Several MSMs of the same shape are reduced in one launch.
constexpr int MSM_MAX_BATCH = 4;
struct MsmReduceBatch {
    const uint32_t* buckets[MSM_MAX_BATCH];
    const uint32_t* counts[MSM_MAX_BATCH];
    int njobs;
};
// Stage 1 of the Row/Col sums: L lanes per sum, each adds its share (cnt/L strided elements) and writes ONE partial — every
// lane does useful additions (a butterfly inside the wave would run at full wave cost with 32, 16, 8 ... lanes active).
// k_msm_fold then reduces the L partials of each sum K at a time.
template <class F> __global__ void __launch_bounds__(256)
k_msm_rowcol(MsmReduceBatch rb, uint32_t W, uint32_t nb, uint32_t rbits, uint32_t cbits, uint32_t L, uint32_t* __restrict__ out) {
    constexpr int PW = 4 * FieldWords<F>::value;
    const uint32_t C = 1u << cbits, R = 1u << rbits;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sub = (uint32_t)(tid & (L - 1));
    const size_t gw = tid / L;                                              // sum index: ((job*W + w)*2 + kind)*C + i
    const size_t n_out = (size_t)rb.njobs * W * 2 * C;
    if (gw >= n_out) return;
    XYZZ<F> acc;
    pt_set_inf(acc);
    const uint32_t i = (uint32_t)(gw & (C - 1)), kind = (uint32_t)(gw >> cbits) & 1u;
    const size_t jw = gw >> (cbits + 1);
    const uint32_t job = (uint32_t)(jw / W), w = (uint32_t)(jw % W);
    const uint32_t* bk = rb.buckets[job];
    const uint32_t* cn = rb.counts[job];
    const uint32_t cnt = kind ? R : ((i < R) ? C : 0u);
    for (uint32_t e = sub; e < cnt; e += L) {
        const size_t g = (size_t)w * nb + (kind ? ((size_t)e << cbits) + i : ((size_t)i << cbits) + e);
        if (cn[g]) {                                                                    // empty buckets were never written
            XYZZ<F> p; pt_load(p, bk + g * PW);
            if constexpr (FieldWords<F>::value <= 12) acc = pt_add_inl(acc, p); else acc = pt_add(acc, p);     // G1: throughput-bound, inline
        }
    }
    pt_store(out + tid * PW, acc);
}
// out[i] = sum_{k<K} in[i*K + k]
template <class F> __global__ void __launch_bounds__(256)
k_msm_fold(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n_out, uint32_t K) {
    constexpr int PW = 4 * FieldWords<F>::value;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    XYZZ<F> acc;
    pt_load(acc, in + i * K * PW);
    for (uint32_t k = 1; k < K; k++) {
        XYZZ<F> p; pt_load(p, in + (i * K + k) * PW);
        if constexpr (FieldWords<F>::value <= 12) acc = pt_add_inl(acc, p); else acc = pt_add(acc, p);
    }
    pt_store(out + i * PW, acc);
}
// ---- row/column sums, one WAVE per sum (no fold launches) ---------------------------------------------------------------
// The 64 lanes of a wave each add a strided share of the sum's buckets (C/64 or R/64 sequential additions), then the wave's 64
// partials are combined by a 6-level tree through LDS inside the same launch. Depth for c = 20: 16 + 6 point additions in ONE
// launch instead of 4 + 3 + 3 + 3 + 3 spread over five (k_msm_rowcol + four k_msm_fold). Fq2 points: the accumulator of every
// lane is parked in LDS and the other operand is streamed coordinate by coordinate (curve.cuh: pt_add_lds) — no scratch frames,
// 2 waves per SIMD; Fq points: accumulator in registers, tree operands through LDS.
template <class F, int T> ZK_DEV void lds_tree_sum(const LdsAcc<F, T>& A, bool& inf, uint32_t* inf_s, uint32_t t, uint32_t group) {
    for (uint32_t d = 1; d < group; d <<= 1) {
        inf_s[t] = inf ? 1u : 0u;
        __syncthreads();
        if ((t & (2 * d - 1)) == 0 && !inf_s[t + d]) {
            const LdsAcc<F, T> Pn{A.base + d};
            pt_add_lds(A, inf, [&](int coord, F& v) { Pn.get(coord, v); });
        }
        __syncthreads();
    }
}
template <class F, int T> ZK_DEV void lds_acc_store(const LdsAcc<F, T>& A, bool inf, uint32_t* dst) {
    constexpr int FW = FieldWords<F>::value;
    if (inf) {
#pragma unroll
        for (int i = 0; i < FW; i++) reinterpret_cast<uint4*>(dst)[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    F v;
#pragma unroll
    for (int cdn = 0; cdn < 4; cdn++) { A.get(cdn, v); f_store(dst + cdn * FW, v); }
}
template <class F> struct MsmRcBlock { static constexpr int value = (FieldWords<F>::value > 12) ? MsmAccumBlock<F>::value : 256; };
template <class F> __global__ void __launch_bounds__(MsmRcBlock<F>::value, (FieldWords<F>::value > 12) ? 2 : 1)
k_msm_rowcol_wave(MsmReduceBatch rb, uint32_t W, uint32_t nb, uint32_t rbits, uint32_t cbits, uint32_t* __restrict__ out) {
    constexpr int FW = FieldWords<F>::value, PW = 4 * FW;
    constexpr bool WIDE = FW > 12;
    constexpr int T = MsmRcBlock<F>::value;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t inf_s[T];
    const uint32_t C = 1u << cbits, R = 1u << rbits;
    const uint32_t t = threadIdx.x, sub = t & 63u;
    const size_t n_out = (size_t)rb.njobs * W * 2 * C;
    // persistent blocks: a grid smaller than the number of sums strides over them. The Fq2 reduction runs on the auxiliary stream
    // underneath the G1 accumulations and its blocks fill a CU's register file (256 VGPRs x 2 waves per SIMD): capped at a quarter of
    // the chip it leaves the other CUs to the main stream instead of stalling it for the whole reduction
    for (size_t blk = blockIdx.x; blk * (T / 64) < n_out; blk += gridDim.x) {
    const size_t gw = blk * (T / 64) + (t >> 6);            // sum index: ((job*W + w)*2 + kind)*C + i
    const bool valid = gw < n_out;
    const uint32_t i = (uint32_t)(gw & (C - 1)), kind = (uint32_t)(gw >> cbits) & 1u;
    const size_t jw = gw >> (cbits + 1);
    const uint32_t job = valid ? (uint32_t)(jw / W) : 0u, w = (uint32_t)(jw % W);
    const uint32_t* bk = rb.buckets[job];
    const uint32_t* cn = rb.counts[job];
    const uint32_t cnt = !valid ? 0u : (kind ? R : ((i < R) ? C : 0u));
    if constexpr (WIDE) {
        const LdsAcc<F, T> A{lds + t};
        bool inf = true;
        for (uint32_t e = sub; e < cnt; e += 64) {
            const size_t g = (size_t)w * nb + (kind ? ((size_t)e << cbits) + i : ((size_t)i << cbits) + e);
            if (cn[g]) {                                                                // empty buckets were never written
                const uint32_t* p = bk + g * PW;
                pt_add_lds(A, inf, [&](int coord, F& v) { f_load(v, p + coord * FW); });
            }
        }
        lds_tree_sum(A, inf, inf_s, t, 64);
        if (valid && sub == 0) lds_acc_store(A, inf, out + gw * PW);
    } else {
        XYZZ<F> acc;
        pt_set_inf(acc);
        for (uint32_t e = sub; e < cnt; e += 64) {
            const size_t g = (size_t)w * nb + (kind ? ((size_t)e << cbits) + i : ((size_t)i << cbits) + e);
            if (cn[g]) { XYZZ<F> p; pt_load(p, bk + g * PW); acc = pt_add_inl(acc, p); }
        }
        for (uint32_t d = 1; d < 64; d <<= 1) {
            pt_store(lds + t * PW, acc);
            __syncthreads();
            if ((t & (2 * d - 1)) == 0) { XYZZ<F> o; pt_load(o, lds + (t + d) * PW); acc = pt_add_inl(acc, o); }
            __syncthreads();
        }
        if (valid && sub == 0) pt_store(out + gw * PW, acc);
    }
    __syncthreads();
    }
}
// k_msm_bitsums for Fq2 points with LDS-parked accumulators (see k_msm_rowcol_wave): one block per (array, k)
template <class F> __global__ void __launch_bounds__(MsmAccumBlock<F>::value, 2)
k_msm_bitsums_lds(const uint32_t* __restrict__ arr, uint32_t C, uint32_t cbits, uint32_t* __restrict__ out) {
    constexpr int FW = FieldWords<F>::value, PW = 4 * FW;
    constexpr int T = MsmAccumBlock<F>::value;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t inf_s[T];
    const uint32_t a = blockIdx.x / (cbits + 1), k = blockIdx.x % (cbits + 1), t = threadIdx.x;
    const LdsAcc<F, T> A{lds + t};
    bool inf = true;
    for (uint32_t i = t; i < C; i += T)
        if (k == cbits || ((i >> k) & 1u)) {
            const uint32_t* p = arr + ((size_t)a * C + i) * PW;
            pt_add_lds(A, inf, [&](int coord, F& v) { f_load(v, p + coord * FW); });
        }
    lds_tree_sum(A, inf, inf_s, t, (uint32_t)T);
    if (t == 0) lds_acc_store(A, inf, out + (size_t)blockIdx.x * PW);
}

// One block of M lanes per M consecutive items of an array of m_per_array points; invariant across levels:
//   weighted = sum_t A_t + scale * sum_t t*X_t,  total = sum_t X_t   (level 0: A absent, scale 1).
template <class F, int M> __global__ void __launch_bounds__(M)
k_msm_wsum(const uint32_t* __restrict__ inA, const uint32_t* __restrict__ inR, uint32_t m_per_window, uint32_t blocks_per_window,
           int log_scale, uint32_t* __restrict__ outA, uint32_t* __restrict__ outR) {
    constexpr int PW = 4 * FieldWords<F>::value;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* sR = lds;
    uint32_t* sA = lds + M * PW;
    const uint32_t w = blockIdx.x / blocks_per_window, blk = blockIdx.x % blocks_per_window, t = threadIdx.x;
    const uint32_t idx = blk * M + t;
    XYZZ<F> myR, myA;
    pt_set_inf(myA);
    if (idx < m_per_window) {
        pt_load(myR, inR + ((size_t)w * m_per_window + idx) * PW);
        if (inA) pt_load(myA, inA + ((size_t)w * m_per_window + idx) * PW);
    } else pt_set_inf(myR);
    pt_store(sR + t * PW, myR);
    __syncthreads();
    const uint32_t act = min((uint32_t)M, m_per_window - blk * M);
    int MA = 1;
    while ((uint32_t)MA < act) MA <<= 1;
    // inclusive suffix scan of X (Hillis-Steele): afterwards lane t holds sum_{u>=t} X_u
#pragma unroll 1
    for (int d = 1; d < MA; d <<= 1) {
        XYZZ<F> o;
        if (t + d < (uint32_t)MA) pt_load(o, sR + (t + d) * PW); else pt_set_inf(o);
        __syncthreads();
        myR = pt_add(myR, o);
        pt_store(sR + t * PW, myR);
        __syncthreads();
    }
    // sum_t t*X_t = sum_{t>=1} suffix_t; tree-sum it in lanes [0,MA) while lanes [MA,2MA) (if present) tree-sum the A's
    XYZZ<F> myX = myR;
    if (t == 0) pt_set_inf(myX);
    pt_store(sR + t * PW, myX);
    pt_store(sA + t * PW, myA);
    __syncthreads();
#pragma unroll 1
    for (int d = MA / 2; d >= 1; d >>= 1) {
        if (t < (uint32_t)d) { XYZZ<F> o; pt_load(o, sR + (t + d) * PW); myX = pt_add(myX, o); }
        if (inA && t < (uint32_t)d) { XYZZ<F> p; pt_load(p, sA + (t + d) * PW); myA = pt_add(myA, p); }
        __syncthreads();
        if (t < (uint32_t)d) { pt_store(sR + t * PW, myX); if (inA) pt_store(sA + t * PW, myA); }
        __syncthreads();
    }
    if (t == 0) {
        for (int k = 0; k < log_scale; k++) myX = pt_dbl(myX);
        XYZZ<F> a = inA ? pt_add(myA, myX) : myX;
        size_t o = (size_t)w * blocks_per_window + blk;
        pt_store(outA + o * PW, a);
        pt_store(outR + o * PW, myR);          // lane 0's suffix sum = total
    }
}

// Alternative to k_msm_wsum when there are few arrays (pre-computed tables: one window): only PLAIN sums on the device,
//   T_k = sum_{t : bit k of t set} X_t  (k < cbits)  and  T_cbits = sum_t X_t,
// one block per (array, k); the host forms sum_t t*X_t = sum_k 2^k T_k by Horner (O(cbits) group operations). Depth on the
// device = C/M sequential additions + log2(M) tree levels, with no doublings in the chain.
template <class F, int M> __global__ void __launch_bounds__(M)
k_msm_bitsums(const uint32_t* __restrict__ arr, uint32_t C, uint32_t cbits, uint32_t* __restrict__ out) {
    constexpr int PW = 4 * FieldWords<F>::value;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t a = blockIdx.x / (cbits + 1), k = blockIdx.x % (cbits + 1), t = threadIdx.x;
    XYZZ<F> acc;
    pt_set_inf(acc);
    for (uint32_t i = t; i < C; i += M)
        if (k == cbits || ((i >> k) & 1u)) { XYZZ<F> p; pt_load(p, arr + ((size_t)a * C + i) * PW); acc = pt_add(acc, p); }
    pt_store(lds + t * PW, acc);
    __syncthreads();
#pragma unroll 1
    for (int d = M / 2; d >= 1; d >>= 1) {
        if (t < (uint32_t)d) { XYZZ<F> o; pt_load(o, lds + (t + d) * PW); acc = pt_add(acc, o); pt_store(lds + t * PW, acc); }
        __syncthreads();
    }
    if (t == 0) pt_store(out + (size_t)blockIdx.x * PW, acc);
}

// one bit per point of a resident base array / table: 1 = point at infinity (mask must be zeroed first)
template <class F> __global__ void k_msm_infmask(const uint32_t* __restrict__ pts, size_t n, uint32_t* __restrict__ mask) {
    constexpr int FW = FieldWords<F>::value;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p;
    pt_load(p, pts + i * 2 * FW);
    if (pt_is_inf(p)) atomicOr(&mask[i >> 5], 1u << (i & 31));
}

// ---- pre-computed window tables for resident bases ------------------------------------------------------------------------
// T[k][i] = 2^(c*k) * P_i in affine form, k < Wd. With the table every signed digit of a scalar lands in ONE set of 2^(c-1)
// buckets, so c can grow (fewer digits per scalar => fewer mixed additions and fewer sort entries) without multiplying the
// bucket-reduction work by the number of windows. One lane per point; one Fermat inversion per table entry (one-off per key).
template <class F> __global__ void __launch_bounds__(256)
k_msm_precompute(const uint32_t* __restrict__ bases, uint32_t n, int c, int Wd, uint32_t* __restrict__ table) {
    constexpr int FW = FieldWords<F>::value;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p;
    pt_load(p, bases + (size_t)i * 2 * FW);
    f_store(table + (size_t)i * 2 * FW, p.x); f_store(table + (size_t)i * 2 * FW + FW, p.y);
    const bool inf = pt_is_inf(p);
    for (int k = 1; k < Wd; k++) {
        uint32_t* dst = table + ((size_t)k * n + i) * 2 * FW;
        if (!inf) {
            XYZZ<F> q = pt_dbl_affine(p);
            for (int d = 1; d < c; d++) q = pt_dbl(q);
            // affine: 1/ZZ = (ZZ/ZZZ)^2 since ZZ^3 = ZZZ^2
            F i3 = f_inv(q.ZZZ), i2 = f_sqr(f_mul(q.ZZ, i3));
            p.x = f_mul(q.X, i2); p.y = f_mul(q.Y, i3);
        }
        f_store(dst, p.x); f_store(dst + FW, p.y);
    }
}

// ---- synthetic base table (zkmi_gen_geometric_bases_dev): P_i = (f*g^i mod r)*G ----------------------------------------
// One lane per point: scalar k_i = f*g^i by square-and-multiply in Fr, then k_i*G by double-and-add (XYZZ), then
// affine by one Fermat inversion. Off the hot path (benchmark/test set-up only).
template <class F, class FrC> __global__ void __launch_bounds__(256)
k_gen_geometric_bases(const uint32_t* __restrict__ gen_affine, uint32_t n, uint64_t f, uint64_t g, uint32_t* __restrict__ out) {
    constexpr int FW = FieldWords<F>::value;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<FrC> fm = fp_zero<FrC>(), gm = fp_zero<FrC>();
    fm.l[0] = (uint32_t)f; fm.l[1] = (uint32_t)(f >> 32); gm.l[0] = (uint32_t)g; gm.l[1] = (uint32_t)(g >> 32);
    fm = fp_to_mont(fm); gm = fp_to_mont(gm);
    Fp<FrC> k = fp_from_mont(fp_mul(fm, fp_pow_u32(gm, i)));
    Affine<F> G; pt_load(G, gen_affine);
    XYZZ<F> acc; pt_set_inf(acc);
    for (int b = 32 * FrC::N - 1; b >= 0; b--) {
        acc = pt_dbl(acc);
        if ((k.l[b >> 5] >> (b & 31)) & 1) pt_madd(acc, G);
    }
    Affine<F> r;
    if (pt_is_inf(acc)) { f_set_zero(r.x); f_set_zero(r.y); }
    else { r.x = f_mul(acc.X, f_inv(acc.ZZ)); r.y = f_mul(acc.Y, f_inv(acc.ZZZ)); }
    f_store(out + (size_t)i * 2 * FW, r.x); f_store(out + (size_t)i * 2 * FW + FW, r.y);
}

// P_i = k_i * G for caller-supplied scalars k_i (32-byte little-endian integers, normal form): bases of synthetic VALID proving keys
// (tests/synth_valid_groth16.py: A_i = u_i(tau) G, ...). One lane per point, double-and-add + one Fermat inversion; set-up only.
template <class F> __global__ void __launch_bounds__(256)
k_gen_scalar_bases(const uint32_t* __restrict__ gen_affine, const uint32_t* __restrict__ scalars, uint32_t n, uint32_t* __restrict__ out) {
    constexpr int FW = FieldWords<F>::value;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8];
#pragma unroll
    for (int w = 0; w < 8; w++) k[w] = scalars[(size_t)i * 8 + w];
    Affine<F> G; pt_load(G, gen_affine);
    XYZZ<F> acc; pt_set_inf(acc);
    for (int b = 255; b >= 0; b--) {
        acc = pt_dbl(acc);
        if ((k[b >> 5] >> (b & 31)) & 1) pt_madd(acc, G);
    }
    Affine<F> r;
    if (pt_is_inf(acc)) { f_set_zero(r.x); f_set_zero(r.y); }
    else { r.x = f_mul(acc.X, f_inv(acc.ZZ)); r.y = f_mul(acc.Y, f_inv(acc.ZZZ)); }
    f_store(out + (size_t)i * 2 * FW, r.x); f_store(out + (size_t)i * 2 * FW + FW, r.y);
}

}  // namespace zkmi
