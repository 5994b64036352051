// snarkjs_amd/csrc/groth16.hip — fused Groth16 prover: the whole of groth16Prove (reference src/groth16_prove.js:28-144)
// between "sections read" and "proof points", on the device.
//
//   buildABC1 (:147-187, a single-threaded JS loop in the reference)  ->  CSR sparse matrix x witness kernel
//   3 x { Fr.ifft -> Fr.batchApplyKey(1, w[power+1]) -> Fr.fft } (:64-76) ->  iNTT + NTT with the coset scale fused on load
//   joinABC (:320-374)                                                ->  element-wise kernel
//   5 x multiExpAffine (:84-101)                                      ->  device Pippenger; A, B1, B2 and C share ONE digit
//                                                                         sort of the witness (C = witness[nPublic+1:])
//   blinding + toAffine (:103-132)                                    ->  host (O(1) group operations)
//
// The zkey is static per circuit: zkmi_groth16_load uploads the five base tables once and converts the coefficient
// section into CSR form (rows = (matrix, constraint)) on the device; a proof then moves only the witness.
#include <stdlib.h>
#include <string.h>
#include <memory>
#include "msm_host.hpp"
#include "zkmi_common.hpp"

namespace zkmi {

// ---- coefficient section -> sliced ELL with split rows (buildABC1, src/groth16_prove.js:147-187) ----------------------------------------
// zkey section 4: u32 nCoef, then nCoef x { u32 matrix, u32 constraint, u32 signal, Fr value (x R^2) } (44-byte records,
// src/zkey_utils.js:108-118 / src/zkey_new.js:320-333). Row id = matrix * domain + constraint: A_T | B_T are one vector of 2 * domain rows.
//
// The rows of a compiled circuit are heavy-tailed: nine in ten hold one term, a few hold 10^3 - 10^5 (Num2Bits, packing, long linear
// combinations). One lane per constraint (r01 - r05) leaves a 10^5-term row to ONE lane — 10^5 dependent gathers and products while the rest of
// the chip idles. Layout built once per key, on the device:
//   * every row is cut into SEGMENTS of at most ABC_SEG terms (an empty row is one segment of length 0: it writes the zero);
//   * the segments are sorted by length, longest first (stable counting sort: per-wave bin counts, one flat scan, ranks), and every 64
//     consecutive ones form a SLICE, as wide as its first = longest segment (SELL-64-sigma with sigma = everything): lanes of a wave run
//     the same trip count, and term k of lane l of a slice lies at (slice_off + k) * 64 + l — the wave reads 2 KB of values and 256 B of
//     signal ids contiguously per step;
//   * a segment of a single-segment row writes its sum straight into A_T | B_T; the segments of a cut row write partial sums, which one wave
//     per such row adds up afterwards (k_abc_long); C_T = A_T * B_T is an element-wise pass (k_abc_mulc).
// Field addition is exact, so the order in which a row's terms meet its segments (atomic cursors below) does not change a single bit.
constexpr int COEF_REC_WORDS = 11;
constexpr uint32_t ABC_SEG = 32;                 // terms per segment: a slice of full width is 32 dependent product-accumulates per lane
constexpr uint32_t ABC_PART = 0x80000000u;       // s_out: the segment writes partial sum (s_out & ~ABC_PART) instead of row s_out
static __global__ void k_coef_count(const uint32_t* __restrict__ raw, uint32_t n_coef, uint32_t domain, uint32_t n_vars, uint32_t* __restrict__ row_cnt, uint32_t* __restrict__ bad) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_coef) return;
    const uint32_t* rec = raw + 1 + (size_t)i * COEF_REC_WORDS;
    uint32_t m = rec[0], c = rec[1], s = rec[2];
    if (m > 1 || c >= domain || s >= n_vars) { atomicAdd(bad, 1u); return; }
    atomicAdd(&row_cnt[m * domain + c], 1u);
}
// exclusive scan of `n` counters by one workgroup (one-off per zkey); out must not alias in
static __global__ void k_scan_u32(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += in[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < blockDim.x; d <<= 1) {
        uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t b = lo; b < hi; b++) { out[b] = run; run += in[b]; }
}
// per row: its number of segments and, for rows cut into several, of partial-sum slots
static __global__ void k_abc_row_segs(const uint32_t* __restrict__ row_cnt, uint32_t rows, uint32_t* __restrict__ nseg, uint32_t* __restrict__ npart) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint32_t c = row_cnt[r], ns = c ? (c + ABC_SEG - 1) / ABC_SEG : 1u;
    nseg[r] = ns;
    npart[r] = ns > 1 ? ns : 0u;
}
// one lane per row: lengths and targets of its segments; cut rows join the list of long rows (row, first partial slot, segments)
static __global__ void k_abc_seg_fill(const uint32_t* __restrict__ row_cnt, const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ part_base, uint32_t rows,
                                      uint32_t* __restrict__ seg_len, uint32_t* __restrict__ seg_out, uint32_t* __restrict__ long_rows, uint32_t* __restrict__ n_long) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint32_t c = row_cnt[r], sb = seg_base[r];
    if (c <= ABC_SEG) { seg_len[sb] = c; seg_out[sb] = r; return; }
    const uint32_t ns = (c + ABC_SEG - 1) / ABC_SEG, pb = part_base[r];
    for (uint32_t j = 0; j < ns; j++) { seg_len[sb + j] = min(ABC_SEG, c - j * ABC_SEG); seg_out[sb + j] = ABC_PART | (pb + j); }
    const uint32_t k = atomicAdd(n_long, 1u);
    long_rows[3 * (size_t)k] = r; long_rows[3 * (size_t)k + 1] = pb; long_rows[3 * (size_t)k + 2] = ns;
}
// Stable counting sort of the segments by length, longest first. One wave per 64 segments (blockDim = 64): the lanes that share a length are found
// by ballots; counts[(ABC_SEG - len) * n_waves + wave] = their number. One flat exclusive scan over that [bin][wave] array is every (bin, wave)'s
// first position in the sorted order; the rank inside the wave is the number of lower lanes of the same length.
static __global__ void __launch_bounds__(64) k_abc_wave_hist(const uint32_t* __restrict__ seg_len, uint32_t n_seg, uint32_t n_waves, uint32_t* __restrict__ counts) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    const bool live = s < n_seg;
    const uint32_t len = live ? seg_len[s] : 0xffffffffu;
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t v = __shfl(len, leader);
        const unsigned long long same = __ballot(live && len == v);
        if ((int)threadIdx.x == leader) counts[(size_t)(ABC_SEG - v) * n_waves + blockIdx.x] = (uint32_t)__popcll(same);
        todo &= ~same;
    }
}
static __global__ void __launch_bounds__(64) k_abc_wave_rank(const uint32_t* __restrict__ seg_len, const uint32_t* __restrict__ seg_out, uint32_t n_seg, uint32_t n_waves,
                                                            const uint32_t* __restrict__ offs, uint32_t* __restrict__ s_len, uint32_t* __restrict__ s_out, uint32_t* __restrict__ pos_of_seg) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    const bool live = s < n_seg;
    const uint32_t len = live ? seg_len[s] : 0xffffffffu;
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t v = __shfl(len, leader);
        const unsigned long long same = __ballot(live && len == v);
        if (live && len == v) {
            const uint32_t pos = offs[(size_t)(ABC_SEG - v) * n_waves + blockIdx.x] + (uint32_t)__popcll(same & ((1ull << threadIdx.x) - 1ull));
            s_len[pos] = len; s_out[pos] = seg_out[s]; pos_of_seg[s] = pos;
        }
        todo &= ~same;
    }
}
static __global__ void k_abc_slice_w(const uint32_t* __restrict__ s_len, uint32_t n_slices, uint32_t* __restrict__ slice_w) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slices) slice_w[i] = s_len[(size_t)i * 64];       // sorted longest first: the slice's first segment is its widest
}
// one lane per coefficient record: its place in the sliced layout
static __global__ void k_abc_sell_fill(const uint32_t* __restrict__ raw, uint32_t n_coef, uint32_t domain, uint32_t n_vars, const uint32_t* __restrict__ seg_base,
                                       const uint32_t* __restrict__ pos_of_seg, const uint32_t* __restrict__ slice_off, uint32_t* __restrict__ cursor,
                                       uint32_t* __restrict__ sell_sig, uint32_t* __restrict__ sell_val) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_coef) return;
    const uint32_t* rec = raw + 1 + (size_t)i * COEF_REC_WORDS;
    uint32_t m = rec[0], c = rec[1], s = rec[2];
    if (m > 1 || c >= domain || s >= n_vars) return;
    const uint32_t row = m * domain + c, rank = atomicAdd(&cursor[row], 1u);
    const uint32_t pos = pos_of_seg[seg_base[row] + rank / ABC_SEG];
    const size_t e = ((size_t)slice_off[pos >> 6] + rank % ABC_SEG) * 64 + (pos & 63u);
    sell_sig[e] = s;
#pragma unroll
    for (int k = 0; k < 8; k++) sell_val[e * 8 + k] = rec[3 + k];
}
// ---- per proof -------------------------------------------------------------------------------------------------------------------------------
// One lane per segment, one wave per slice: sum of coef * w[signal] over the segment's terms. coef is stored x R^2 and the witness is in normal
// form, so the Montgomery product is (coef * w) x R (buildABC1 :166-178).
template <class C> __global__ void __launch_bounds__(256)
k_abc_sell(const uint32_t* __restrict__ s_len, const uint32_t* __restrict__ s_out, const uint32_t* __restrict__ slice_off, uint32_t n_seg, const uint32_t* __restrict__ sell_sig,
           const uint32_t* __restrict__ sell_val, const uint32_t* __restrict__ witness, uint32_t* __restrict__ AB, uint32_t* __restrict__ part) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x, slice = p >> 6;
    if ((size_t)slice * 64 >= n_seg) return;                         // the whole wave
    const uint32_t width = __builtin_amdgcn_readfirstlane(s_len[(size_t)slice * 64]);
    const uint32_t len = p < n_seg ? s_len[p] : 0u;
    size_t e = (size_t)__builtin_amdgcn_readfirstlane(slice_off[slice]) * 64 + (p & 63u);
    Fp<C> acc = fp_zero<C>();
    for (uint32_t k = 0; k < width; k++, e += 64) {
        if (k < len) {
            const Fp<C> v = fp_load<C>(sell_val + e * 8);
            const Fp<C> w = fp_load<C>(witness + (size_t)sell_sig[e] * 8);
            acc = fp_add(acc, fp_mul(v, w));
        }
    }
    if (p < n_seg) {
        const uint32_t o = s_out[p];
        fp_store<C>((o & ABC_PART) ? part + (size_t)(o & ~ABC_PART) * 8 : AB + (size_t)o * 8, acc);
    }
}
// one wave per cut row: adds its partial sums
template <class C> __global__ void __launch_bounds__(256)
k_abc_long(const uint32_t* __restrict__ long_rows, uint32_t n_long, const uint32_t* __restrict__ part, uint32_t* __restrict__ AB) {
    const uint32_t wv = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (wv >= n_long) return;
    const uint32_t r = long_rows[3 * (size_t)wv], pb = long_rows[3 * (size_t)wv + 1], ns = long_rows[3 * (size_t)wv + 2];
    Fp<C> acc = fp_zero<C>();
    for (uint32_t j = lane; j < ns; j += 64) acc = fp_add(acc, fp_load<C>(part + (size_t)(pb + j) * 8));
    for (int off = 32; off; off >>= 1) {
        Fp<C> o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = __shfl_down(acc.l[i], off);
        acc = fp_add(acc, o);
    }
    if (lane == 0) fp_store<C>(AB + (size_t)r * 8, acc);
}
// C_T[c] = A_T[c] * B_T[c] (buildABC1 :180-184)
template <class C> __global__ void __launch_bounds__(256)
k_abc_mulc(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, uint32_t* __restrict__ Cc, uint32_t domain) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= domain) return;
    fp_store<C>(Cc + (size_t)c * 8, fp_mul(fp_load<C>(A + (size_t)c * 8), fp_load<C>(B + (size_t)c * 8)));
}

// ---- paged host sections (zkmi_groth16_zkey_paged) -----------------------------------------------------------------------------------------------
static size_t pg_total(const zkmi_pages& pg) { size_t t = 0; for (int i = 0; i < pg.n_pages; i++) t += pg.len[i]; return t; }
// bytes [off, off + len) of a paged host buffer -> device (stream-ordered) / -> host. A page with a NULL pointer is a GAP: bytes the caller did not
// read (a shard loader reads only its slice of a section); a range that needs one fails.
template <class F> static int pg_walk(const zkmi_pages& pg, size_t off, size_t len, F&& f) {
    size_t start = 0, done = 0;
    for (int i = 0; i < pg.n_pages && done < len; i++) {
        const size_t pl = pg.len[i], lo = off + done;
        if (lo < start + pl) {
            const size_t in = lo - start, k = std::min(pl - in, len - done);
            if (!pg.ptr[i]) return fail(ZKMI_ERR_INVALID, "groth16: a needed byte range of a zkey section lies in a gap page (not provided by the caller)");
            ZK_TRY(f(pg.ptr[i] + in, done, k));
            done += k;
        }
        start += pl;
    }
    if (done < len) return fail(ZKMI_ERR_INVALID, "groth16: a zkey section is shorter than the range read from it");
    return ZKMI_OK;
}
static int pg_upload(const zkmi_pages& pg, size_t off, size_t len, void* d_dst, hipStream_t st) {
    return pg_walk(pg, off, len, [&](const uint8_t* src, size_t at, size_t k) -> int {
        ZK_HIP(hipMemcpyAsync((uint8_t*)d_dst + at, src, k, hipMemcpyHostToDevice, st));
        return ZKMI_OK;
    });
}
static int pg_read(const zkmi_pages& pg, size_t off, size_t len, uint8_t* dst) {
    return pg_walk(pg, off, len, [&](const uint8_t* src, size_t at, size_t k) -> int { memcpy(dst + at, src, k); return ZKMI_OK; });
}

// ---- resident proving key ---------------------------------------------------------------------------------------------
enum { ST_BUILD = 0, ST_NTT, ST_JOIN, ST_SORT_W, ST_MSM_B2, ST_MSM_B1, ST_MSM_A, ST_MSM_C, ST_SORT_H, ST_MSM_H, ST_REDUCE, ST_COUNT };

struct G16Key {
    int curve = 0;
    uint32_t n_vars = 0, n_public = 0, domain = 0, power = 0, n_coef = 0;
    void *bA = nullptr, *bB1 = nullptr, *bB2 = nullptr, *bC = nullptr, *bH = nullptr;   // base tables; with pre-computed windows: T[k][i] = 2^(c*k) P_i
    uint32_t* drop_b = nullptr;       // scalars whose B1 AND B2 bases are both at infinity: left out of the sort that feeds B1/B2
    double b_density = 1.0;           // fraction of witness entries that survive drop_b
    uint32_t* mask[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // infinity bitmaps of bA, bB1, bB2, bC, bH (one bit per table entry)
    int cw = 0, ch = 0;               // window width of the witness-side / H-side tables (0 = plain bases, no pre-computation)
    uint32_t c_skip = 0;              // index offset of the C bases (0 when the C table is padded to nVars entries)
    // shard of the key held by this device (multi-GPU: MSMs split by base-index range): witness-side bases of the
    // variables [v_lo, v_lo + v_cnt) and H bases [h_lo, h_lo + h_cnt); the full key has v_lo = h_lo = 0, v_cnt = nVars, h_cnt = n
    uint32_t v_lo = 0, v_cnt = 0, h_lo = 0, h_cnt = 0;
    bool full() const { return v_lo == 0 && h_lo == 0 && v_cnt == n_vars && h_cnt == domain; }
    // the coefficient section as sliced ELL with split rows (above): lengths / targets of the sorted segments, slice offsets (units of 64 terms),
    // signal ids and values of the terms, the list of cut rows (row, first partial slot, segments)
    uint32_t *s_len = nullptr, *s_out = nullptr, *slice_off = nullptr, *sell_sig = nullptr, *sell_val = nullptr, *long_rows = nullptr;
    uint32_t n_seg = 0, n_long = 0, n_part = 0;
    size_t sell_terms = 0;            // padded terms held (64 x the sum of the slice widths)
    // per pipeline slot (two proofs may be in flight, zkmi_groth16_submit_dev): work buffers, stage events, the MSM jobs of the
    // proof in flight. Slot 1 is allocated on first use.
    struct Work {
        uint32_t *w = nullptr, *A = nullptr, *B = nullptr, *C = nullptr, *T = nullptr, *part = nullptr;     // part: partial sums of the cut rows of buildABC
        hipEvent_t ev[ST_COUNT + 1] = {};
        MsmJob job[5];
        bool in_flight = false, ov = false;
        // the witness-side half of a proof has been enqueued (zkmi_groth16_sums_w_dev) and waits for its H half: bucket shape of the
        // witness plan (the H accumulation merges into C's buckets when the shapes agree)
        bool w_enqueued = false;
        int pl_W = 0;
        uint32_t pl_nb = 0;
    } wk[2];
    std::vector<uint8_t> vk_alpha_1, vk_beta_1, vk_beta_2, vk_delta_1, vk_delta_2;
    mutable std::vector<uint8_t> fb_delta1, fb_delta2;   // host fixed-base tables of delta (g16_finish), built on first use
    double stage_ms[ST_COUNT] = {};
    void release() {
        void** ptrs[] = {&bA, &bB1, &bB2, &bC, &bH, (void**)&s_len, (void**)&s_out, (void**)&slice_off, (void**)&sell_sig, (void**)&sell_val, (void**)&long_rows,
                         (void**)&mask[0], (void**)&mask[1], (void**)&mask[2],
                         (void**)&mask[3], (void**)&mask[4], (void**)&drop_b, (void**)&wk[0].w, (void**)&wk[0].A, (void**)&wk[0].T, (void**)&wk[0].part,
                         (void**)&wk[1].w, (void**)&wk[1].A, (void**)&wk[1].T, (void**)&wk[1].part};        // B and C live inside the A allocation
        for (void* t : {bA, bB1, bB2, bC, bH}) if (t) msm_table_forget_r29(t);
        for (void** p : ptrs) if (*p) { (void)hipFree(*p); *p = nullptr; }
        for (auto& wkk : wk) for (auto& e : wkk.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    }
    ~G16Key() { release(); }          // every early return of g16_load (ZK_TRY / ZK_HIP after make_unique) frees what was allocated so far
    G16Key() = default;
    G16Key(const G16Key&) = delete;
    G16Key& operator=(const G16Key&) = delete;
};
// frees a temporary device allocation on every exit path
struct DevTmp {
    void* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
};

static int g16_work_alloc(G16Key& K, int slot) {
    G16Key::Work& W = K.wk[slot];
    if (W.A) return ZKMI_OK;
    ZK_TRY(dev_alloc_big((void**)&W.w, (size_t)K.n_vars * 32));
    // A | B | C contiguous and T three transforms long: the three chains run as batched NTT launches (ntt_dev_batch_dispatch)
    ZK_TRY(dev_alloc_big((void**)&W.A, (size_t)K.domain * 32 * 3));
    W.B = W.A + (size_t)K.domain * 8; W.C = W.B + (size_t)K.domain * 8;
    ZK_TRY(dev_alloc_big((void**)&W.T, (size_t)K.domain * 32 * 3));
    ZK_TRY(dev_alloc_big((void**)&W.part, std::max<size_t>(K.n_part, 1) * 32));
    for (auto& e : W.ev) ZK_HIP(hipEventCreate(&e));
    return ZKMI_OK;
}
// The resident form of the coefficient section (layout above), built on the device from the raw records
static int g16_build_sell(G16Key& K, const zkmi_pages& coeffs, size_t coeffs_len, hipStream_t st) {
    const uint32_t n = K.domain, n_coef = K.n_coef;
    const size_t rows = 2 * (size_t)n;
    DevTmp raw_t, cnt_t, nseg_t, npart_t, segb_t, partb_t, seglen_t, segout_t, counts_t, offs_t, pos_t, slw_t, misc_t;
    ZK_HIP(hipMalloc(&raw_t.p, coeffs_len + 16));
    ZK_TRY(pg_upload(coeffs, 0, coeffs_len, raw_t.p, st));
    const uint32_t* raw = (const uint32_t*)raw_t.p;
    for (DevTmp* t : {&cnt_t, &nseg_t, &npart_t, &segb_t, &partb_t}) ZK_HIP(hipMalloc(&t->p, rows * 4 + 16));
    ZK_HIP(hipMalloc(&misc_t.p, 64));                                   // [0] bad records, [1] cut rows
    uint32_t *row_cnt = (uint32_t*)cnt_t.p, *nseg = (uint32_t*)nseg_t.p, *npart = (uint32_t*)npart_t.p, *seg_base = (uint32_t*)segb_t.p, *part_base = (uint32_t*)partb_t.p,
             *misc = (uint32_t*)misc_t.p;
    ZK_HIP(hipMemsetAsync(row_cnt, 0, rows * 4, st)); ZK_HIP(hipMemsetAsync(misc, 0, 64, st));
    const unsigned cblocks = (n_coef + 255) / 256, rblocks = (unsigned)((rows + 255) / 256);
    if (n_coef) hipLaunchKernelGGL(k_coef_count, dim3(cblocks), dim3(256), 0, st, raw, n_coef, n, K.n_vars, row_cnt, misc);
    hipLaunchKernelGGL(k_abc_row_segs, dim3(rblocks), dim3(256), 0, st, row_cnt, (uint32_t)rows, nseg, npart);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, nseg, seg_base, (uint32_t)rows);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, npart, part_base, (uint32_t)rows);
    uint32_t tail[5] = {0, 0, 0, 0, 0};                                 // bad, last seg_base, last nseg, last part_base, last npart
    ZK_HIP(hipMemcpyAsync(&tail[0], misc, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipMemcpyAsync(&tail[1], seg_base + rows - 1, 4, hipMemcpyDeviceToHost, st)); ZK_HIP(hipMemcpyAsync(&tail[2], nseg + rows - 1, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipMemcpyAsync(&tail[3], part_base + rows - 1, 4, hipMemcpyDeviceToHost, st)); ZK_HIP(hipMemcpyAsync(&tail[4], npart + rows - 1, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
    if (tail[0]) return fail(ZKMI_ERR_INVALID, "groth16: coefficient record out of range (matrix > 1, constraint >= domain or signal >= nVars)");
    const size_t n_seg = (size_t)tail[1] + tail[2], n_part = (size_t)tail[3] + tail[4];
    if (n_seg >= 0x7fffffffu || n_part >= 0x7fffffffu) return fail(ZKMI_ERR_UNSUPPORTED, "groth16: coefficient section too large for the 31-bit segment indices");
    K.n_seg = (uint32_t)n_seg; K.n_part = (uint32_t)n_part;
    const uint32_t n_waves = (uint32_t)((n_seg + 63) / 64);             // = slices
    ZK_HIP(hipMalloc(&seglen_t.p, n_seg * 4 + 16)); ZK_HIP(hipMalloc(&segout_t.p, n_seg * 4 + 16)); ZK_HIP(hipMalloc(&pos_t.p, n_seg * 4 + 16));
    const size_t n_counts = (size_t)(ABC_SEG + 1) * n_waves;
    if (n_counts >= 0xffffffffu) return fail(ZKMI_ERR_UNSUPPORTED, "groth16: coefficient section too large for the segment sort");
    ZK_HIP(hipMalloc(&counts_t.p, n_counts * 4 + 16)); ZK_HIP(hipMalloc(&offs_t.p, n_counts * 4 + 16)); ZK_HIP(hipMalloc(&slw_t.p, (size_t)n_waves * 4 + 16));
    ZK_HIP(hipMalloc((void**)&K.long_rows, (size_t)(n_coef / (ABC_SEG + 1) + 1) * 12));
    ZK_TRY(dev_alloc_big((void**)&K.s_len, n_seg * 4 + 16)); ZK_TRY(dev_alloc_big((void**)&K.s_out, n_seg * 4 + 16));
    ZK_HIP(hipMalloc((void**)&K.slice_off, (size_t)n_waves * 4 + 16));
    uint32_t *seg_len = (uint32_t*)seglen_t.p, *seg_out = (uint32_t*)segout_t.p, *pos_of_seg = (uint32_t*)pos_t.p, *counts = (uint32_t*)counts_t.p, *offs = (uint32_t*)offs_t.p,
             *slice_w = (uint32_t*)slw_t.p;
    hipLaunchKernelGGL(k_abc_seg_fill, dim3(rblocks), dim3(256), 0, st, row_cnt, seg_base, part_base, (uint32_t)rows, seg_len, seg_out, K.long_rows, misc + 1);
    ZK_HIP(hipMemsetAsync(counts, 0, n_counts * 4, st));
    hipLaunchKernelGGL(k_abc_wave_hist, dim3(n_waves), dim3(64), 0, st, seg_len, (uint32_t)n_seg, n_waves, counts);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, counts, offs, (uint32_t)n_counts);
    hipLaunchKernelGGL(k_abc_wave_rank, dim3(n_waves), dim3(64), 0, st, seg_len, seg_out, (uint32_t)n_seg, n_waves, offs, K.s_len, K.s_out, pos_of_seg);
    hipLaunchKernelGGL(k_abc_slice_w, dim3((n_waves + 255) / 256), dim3(256), 0, st, K.s_len, n_waves, slice_w);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, slice_w, K.slice_off, n_waves);
    uint32_t t2[4] = {0, 0, 0, 0};                                      // cut rows, last slice_off, last slice_w, longest segment count is not needed
    ZK_HIP(hipMemcpyAsync(&t2[0], misc + 1, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipMemcpyAsync(&t2[1], K.slice_off + n_waves - 1, 4, hipMemcpyDeviceToHost, st)); ZK_HIP(hipMemcpyAsync(&t2[2], slice_w + n_waves - 1, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
    K.n_long = t2[0];
    K.sell_terms = ((size_t)t2[1] + t2[2]) * 64;
    ZK_TRY(dev_alloc_big((void**)&K.sell_sig, std::max<size_t>(K.sell_terms, 1) * 4)); ZK_TRY(dev_alloc_big((void**)&K.sell_val, std::max<size_t>(K.sell_terms, 1) * 32));
    ZK_HIP(hipMemsetAsync(row_cnt, 0, rows * 4, st));                   // re-used as the per-row cursors of the fill
    if (n_coef) hipLaunchKernelGGL(k_abc_sell_fill, dim3(cblocks), dim3(256), 0, st, raw, n_coef, n, K.n_vars, seg_base, pos_of_seg, K.slice_off, row_cnt, K.sell_sig, K.sell_val);
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
// buildABC1 of one proof into the slot's A | B | C
template <class FrC> static void g16_build_abc(const G16Key& K, const G16Key::Work& Wk, const uint32_t* w, hipStream_t st) {
    const uint32_t n = K.domain;
    hipLaunchKernelGGL((k_abc_sell<FrC>), dim3((K.n_seg + 255) / 256), dim3(256), 0, st, K.s_len, K.s_out, K.slice_off, K.n_seg, K.sell_sig, K.sell_val, w, Wk.A, Wk.part);
    if (K.n_long) hipLaunchKernelGGL((k_abc_long<FrC>), dim3((K.n_long + 3) / 4), dim3(256), 0, st, K.long_rows, K.n_long, Wk.part, Wk.A);
    hipLaunchKernelGGL((k_abc_mulc<FrC>), dim3((n + 255) / 256), dim3(256), 0, st, Wk.A, Wk.B, Wk.C, n);
}

static int g16_load(const zkmi_groth16_zkey_paged* zk, uint64_t key, uint32_t v_lo, uint32_t v_hi, uint32_t h_lo, uint32_t h_hi) {
    Ctx& cx = ctx();
    hipStream_t st = cx.stream;
    if (zk->curve != ZKMI_CURVE_BN128 && zk->curve != ZKMI_CURVE_BLS12381) return fail(ZKMI_ERR_INVALID, "unknown curve");
    const uint32_t n = zk->domain_size;
    if (n == 0 || (n & (n - 1))) return fail(ZKMI_ERR_INVALID, "groth16: domain size must be a power of two");
    if (n > (1u << 28)) return fail(ZKMI_ERR_UNSUPPORTED, "groth16: domain size beyond 2^28");
    if (zk->n_vars <= zk->n_public) return fail(ZKMI_ERR_INVALID, "groth16: nVars must exceed nPublic");
    for (const zkmi_pages* pg : {&zk->coeffs, &zk->bases_a, &zk->bases_b1, &zk->bases_b2, &zk->bases_c, &zk->bases_h})
        if (pg->n_pages < 0 || (pg->n_pages > 0 && (!pg->ptr || !pg->len))) return fail(ZKMI_ERR_INVALID, "groth16: null section pointer");
    const size_t coeffs_len = pg_total(zk->coeffs);
    if (coeffs_len < 4 || (coeffs_len - 4) % 44) return fail(ZKMI_ERR_INVALID, "groth16: malformed coefficient section");
    if ((coeffs_len - 4) / 44 > 0xffffffffull) return fail(ZKMI_ERR_INVALID, "groth16: malformed coefficient section");
    auto K = std::make_unique<G16Key>();
    K->curve = zk->curve; K->n_vars = zk->n_vars; K->n_public = zk->n_public; K->domain = n;
    K->power = (uint32_t)ilog2_sz(n);
    K->n_coef = (uint32_t)((coeffs_len - 4) / 44);                // buildABC1: nCoef = (byteLength-4)/sCoef (:149-150)
    const size_t q = n8q_of(zk->curve), g1 = 2 * q, g2 = 4 * q;
    {   // section lengths against the header (src/zkey_utils.js:183-205: nVars points in sections 5-7, nVars-nPublic-1 in 8, domainSize in 9)
        const size_t nv = zk->n_vars, nc = nv - zk->n_public - 1;
        if (!zk->coeffs.n_pages || !zk->bases_a.n_pages || !zk->bases_b1.n_pages || !zk->bases_b2.n_pages || (nc && !zk->bases_c.n_pages) || !zk->bases_h.n_pages ||
            !zk->vk_alpha_1 || !zk->vk_beta_1 || !zk->vk_beta_2 || !zk->vk_delta_1 || !zk->vk_delta_2)
            return fail(ZKMI_ERR_INVALID, "groth16: null section pointer");
        if (pg_total(zk->bases_a) < nv * g1 || pg_total(zk->bases_b1) < nv * g1 || pg_total(zk->bases_b2) < nv * g2 || pg_total(zk->bases_c) < nc * g1 ||
            pg_total(zk->bases_h) < (size_t)n * g1)
            return fail(ZKMI_ERR_INVALID, "groth16: a base section is shorter than nVars / nPublic / domainSize of the header require");
        uint32_t n_coef_hdr = 0;
        ZK_TRY(pg_read(zk->coeffs, 0, 4, (uint8_t*)&n_coef_hdr));
        if ((size_t)n_coef_hdr != (coeffs_len - 4) / 44) return fail(ZKMI_ERR_INVALID, "groth16: coefficient count does not match the section length");
    }
    if (v_lo >= v_hi || v_hi > zk->n_vars || h_lo >= h_hi || h_hi > n) return fail(ZKMI_ERR_INVALID, "groth16: empty or out-of-range key shard");
    K->v_lo = v_lo; K->v_cnt = v_hi - v_lo; K->h_lo = h_lo; K->h_cnt = h_hi - h_lo;
    // m = witness-side bases held here; C pairs with witness[nPublic+1:] (:97): c_front = entries of this range that precede it
    const size_t m = K->v_cnt, first_c = (size_t)zk->n_public + 1;
    const size_t c_front = first_c > v_lo ? std::min<size_t>(first_c - v_lo, m) : 0, mc = m - c_front;
    const size_t c_src0 = (v_lo > first_c ? v_lo - first_c : 0);
    // Pre-computed window tables (msm.cuh: k_msm_precompute): the zkey is static, so every base set is expanded once into
    // T[k][i] = 2^(c*k) * P_i. ZKMI_PRECOMP=0 keeps the plain bases (per-window bucket sets); ZKMI_PRECOMP=<c> forces c.
    const char* pe = getenv("ZKMI_PRECOMP");
    const int pc = pe ? atoi(pe) : -1;
    K->cw = pc == 0 ? 0 : (pc > 0 ? pc : msm_precomp_c(m));
    K->ch = pc == 0 ? 0 : (pc > 0 ? pc : msm_precomp_c(K->h_cnt));
    // src: the section's pages; first: index of its first point wanted here (only that byte range is read: the other pages may be gaps)
    auto put_table = [&](void** dst, uint32_t** mask, const zkmi_pages& src, size_t first, size_t cnt, size_t pad_front, int group, int c) -> int {
        const size_t pb = group == 1 ? g1 : g2, tot = cnt + pad_front;
        DevTmp tmp;
        ZK_TRY(dev_alloc_big(&tmp.p, tot * pb ? tot * pb : 16));
        void* raw = tmp.p;
        if (pad_front) ZK_HIP(hipMemsetAsync(raw, 0, pad_front * pb, st));             // all-zero bytes = point at infinity
        if (cnt) ZK_TRY(pg_upload(src, first * pb, cnt * pb, (uint8_t*)raw + pad_front * pb, st));
        const int Wd = c ? msm_digits(32, c) : 1;
        if (!c) { *dst = raw; tmp.p = nullptr; }
        else {
            ZK_TRY(dev_alloc_big(dst, (size_t)Wd * tot * pb));
            ZK_TRY(msm_precompute_dispatch(zk->curve, group, raw, tot, c, Wd, *dst));
        }
        ZK_HIP(hipMalloc((void**)mask, (((size_t)Wd * tot + 31) / 32) * 4 + 16));
        ZK_TRY(msm_infmask_dispatch(zk->curve, group, *dst, (size_t)Wd * tot, *mask));
        if (c) ZK_TRY(msm_table_to_r29(zk->curve, group, *dst, (size_t)Wd * tot, *mask));      // R'-form tables where the 29-bit path exists
        ZK_HIP(hipStreamSynchronize(st));
        return ZKMI_OK;                                    // ~DevTmp frees the plain copy when a table was built from it
    };
    ZK_TRY(put_table(&K->bA, &K->mask[0], zk->bases_a, v_lo, m, 0, 1, K->cw));
    ZK_TRY(put_table(&K->bB1, &K->mask[1], zk->bases_b1, v_lo, m, 0, 1, K->cw));
    ZK_TRY(put_table(&K->bB2, &K->mask[2], zk->bases_b2, v_lo, m, 0, 2, K->cw));
    // with tables C is padded in front so that it shares the witness indices of this range
    ZK_TRY(put_table(&K->bC, &K->mask[3], zk->bases_c, c_src0, mc, K->cw ? c_front : 0, 1, K->cw));
    K->c_skip = K->cw ? 0 : (uint32_t)c_front;
    ZK_TRY(put_table(&K->bH, &K->mask[4], zk->bases_h, h_lo, (size_t)K->h_cnt, 0, 1, K->ch));
    {   // B is sparse in real circuits (a signal absent from the B matrix has the point at infinity in BOTH B1 and B2, section
        // layout src/zkey_utils.js:183-193): those witness entries are dropped from the digit sort that feeds the B MSMs
        const size_t words = (m + 31) / 32;
        std::vector<uint32_t> h1(words), h2(words);
        ZK_HIP(hipMemcpyAsync(h1.data(), K->mask[1], words * 4, hipMemcpyDeviceToHost, st));
        ZK_HIP(hipMemcpyAsync(h2.data(), K->mask[2], words * 4, hipMemcpyDeviceToHost, st));
        ZK_HIP(hipStreamSynchronize(st));
        size_t dropped = 0;
        for (size_t i = 0; i < words; i++) { h1[i] &= h2[i]; dropped += (size_t)__builtin_popcount(h1[i]); }
        K->b_density = 1.0 - (double)dropped / (double)m;
        ZK_HIP(hipMalloc((void**)&K->drop_b, words * 4 + 16));
        ZK_HIP(hipMemcpyAsync(K->drop_b, h1.data(), words * 4, hipMemcpyHostToDevice, st));
        ZK_HIP(hipStreamSynchronize(st));
    }
    K->vk_alpha_1.assign(zk->vk_alpha_1, zk->vk_alpha_1 + g1); K->vk_beta_1.assign(zk->vk_beta_1, zk->vk_beta_1 + g1);
    K->vk_beta_2.assign(zk->vk_beta_2, zk->vk_beta_2 + g2); K->vk_delta_1.assign(zk->vk_delta_1, zk->vk_delta_1 + g1);
    K->vk_delta_2.assign(zk->vk_delta_2, zk->vk_delta_2 + g2);
    ZK_TRY(g16_build_sell(*K, zk->coeffs, coeffs_len, st));
    ZK_TRY(g16_work_alloc(*K, 0));
    auto it = cx.groth16.find(key);
    if (it != cx.groth16.end()) delete (G16Key*)it->second;
    cx.groth16[key] = K.release();
    return ZKMI_OK;
}
// the flat descriptor as a paged one of single pages (the arrays live in `hold`)
struct FlatAsPaged {
    const uint8_t* ptr[6];
    size_t len[6];
    zkmi_groth16_zkey_paged pz;
    explicit FlatAsPaged(const zkmi_groth16_zkey& z) {
        const uint8_t* p[6] = {z.coeffs, z.bases_a, z.bases_b1, z.bases_b2, z.bases_c, z.bases_h};
        const size_t l[6] = {z.coeffs_len, z.bases_a_len, z.bases_b1_len, z.bases_b2_len, z.bases_c_len, z.bases_h_len};
        zkmi_pages* dst[6] = {&pz.coeffs, &pz.bases_a, &pz.bases_b1, &pz.bases_b2, &pz.bases_c, &pz.bases_h};
        for (int i = 0; i < 6; i++) { ptr[i] = p[i]; len[i] = l[i]; dst[i]->ptr = &ptr[i]; dst[i]->len = &len[i]; dst[i]->n_pages = p[i] ? 1 : 0; }
        pz.curve = z.curve; pz.n_vars = z.n_vars; pz.n_public = z.n_public; pz.domain_size = z.domain_size;
        pz.vk_alpha_1 = z.vk_alpha_1; pz.vk_beta_1 = z.vk_beta_1; pz.vk_beta_2 = z.vk_beta_2; pz.vk_delta_1 = z.vk_delta_1; pz.vk_delta_2 = z.vk_delta_2;
    }
    FlatAsPaged(const FlatAsPaged&) = delete;
};
static int g16_load(const zkmi_groth16_zkey* zk, uint64_t key, uint32_t v_lo, uint32_t v_hi, uint32_t h_lo, uint32_t h_hi) {
    FlatAsPaged f(*zk);
    return g16_load(&f.pz, key, v_lo, v_hi, h_lo, h_hi);
}

// ---- host epilogue: blinding + toAffine (src/groth16_prove.js:103-132) -------------------------------------------------
// delta_1 is multiplied by r, s and -rs and delta_2 by s in every proof (:107-120): 8-bit fixed-base tables T[w][d] = d 2^(8w) delta
// (32 x 255 Jacobian points, built once per key) turn each of those 256-bit scalar multiplications into 32 point additions.
template <class CV> static void fixed_base_build(const CV& cv, const typename CV::P& base, std::vector<uint8_t>& blob) {
    typedef typename CV::P P;
    blob.resize((size_t)32 * 255 * sizeof(P));
    P* t = reinterpret_cast<P*>(blob.data());
    P b = base;
    for (int w = 0; w < 32; w++) {
        t[w * 255] = b;
        for (int d = 1; d < 255; d++) t[w * 255 + d] = cv.add(t[w * 255 + d - 1], b);
        for (int k = 0; k < 8; k++) b = cv.dbl(b);
    }
}
template <class CV> static typename CV::P fixed_base_mul(const CV& cv, const std::vector<uint8_t>& blob, const uint8_t* k32) {
    typedef typename CV::P P;
    const P* t = reinterpret_cast<const P*>(blob.data());
    P acc = cv.zero();
    for (int w = 0; w < 32; w++) if (k32[w]) acc = cv.add(acc, t[w * 255 + k32[w] - 1]);
    return acc;
}
template <class G1F, class G2F, class FrC>
static void g16_finish(const G16Key& K, const uint8_t* jA, const uint8_t* jB1, const uint8_t* jB2, const uint8_t* jC, const uint8_t* jH,
                       const uint8_t* r_mont, const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c) {
    typedef typename HostOf<G1F>::FT F1;
    typedef typename HostOf<G2F>::FT F2;
    host::HCurve<F1> c1{HostOf<G1F>::make()};
    host::HCurve<F2> c2{HostOf<G2F>::make()};
    const host::HField<4> Fr = host::HField<4>::from_cfg<FrC>();
    constexpr int B1 = 4 * FieldWords<G1F>::value, B2 = 4 * FieldWords<G2F>::value;   // bytes per coordinate
    auto ld1 = [&](const uint8_t* p) { typename host::HCurve<F1>::P r; memcpy(&r.X, p, B1); memcpy(&r.Y, p + B1, B1); memcpy(&r.Z, p + 2 * B1, B1); return r; };
    auto ld2 = [&](const uint8_t* p) { typename host::HCurve<F2>::P r; memcpy(&r.X, p, B2); memcpy(&r.Y, p + B2, B2); memcpy(&r.Z, p + 2 * B2, B2); return r; };
    auto af1 = [&](const std::vector<uint8_t>& v) { typename F1::E x, y; memcpy(&x, v.data(), B1); memcpy(&y, v.data() + B1, B1); return c1.from_affine(x, y); };
    auto af2 = [&](const std::vector<uint8_t>& v) { typename F2::E x, y; memcpy(&x, v.data(), B2); memcpy(&y, v.data() + B2, B2); return c2.from_affine(x, y); };
    host::HFp<4> r, s;
    memcpy(r.v, r_mont, 32); memcpy(s.v, s_mont, 32);
    // timesFr(P, k): k is a Montgomery Fr element; the scalar is its normal form
    auto bits = [&](const host::HFp<4>& k_mont, uint8_t* out) { host::HFp<4> k = Fr.from_mont(k_mont); memcpy(out, k.v, 32); };
    uint8_t rb[32], sb[32], rsb[32];
    bits(r, rb); bits(s, sb); bits(Fr.neg(Fr.mul(r, s)), rsb);
    auto alpha1 = af1(K.vk_alpha_1), beta1 = af1(K.vk_beta_1), delta1 = af1(K.vk_delta_1);
    auto beta2 = af2(K.vk_beta_2), delta2 = af2(K.vk_delta_2);
    if (K.fb_delta1.empty()) { fixed_base_build(c1, delta1, K.fb_delta1); fixed_base_build(c2, delta2, K.fb_delta2); }
    auto pa = c1.add(c1.add(ld1(jA), alpha1), fixed_base_mul(c1, K.fb_delta1, rb));            // :106-107
    auto pb = c2.add(c2.add(ld2(jB2), beta2), fixed_base_mul(c2, K.fb_delta2, sb));            // :109-110
    auto pb1 = c1.add(c1.add(ld1(jB1), beta1), fixed_base_mul(c1, K.fb_delta1, sb));           // :112-113
    auto pc = c1.add(ld1(jC), ld1(jH));                                                        // :115
    {   // s*pi_a + r*pib1 (:118-119) in one double-and-add pass (Shamir): one set of doublings for both scalars
        const auto both = c1.add(pa, pb1);
        auto acc = c1.zero();
        for (int i = 255; i >= 0; i--) {
            acc = c1.dbl(acc);
            const int bs = (sb[i / 8] >> (i % 8)) & 1, br = (rb[i / 8] >> (i % 8)) & 1;
            if (bs && br) acc = c1.add(acc, both);
            else if (bs) acc = c1.add(acc, pa);
            else if (br) acc = c1.add(acc, pb1);
        }
        pc = c1.add(pc, acc);
    }
    pc = c1.add(pc, fixed_base_mul(c1, K.fb_delta1, rsb));                                     // :120
    typename F1::E x1, y1; typename F2::E x2, y2;
    c1.to_affine(pa, x1, y1); memcpy(pi_a, &x1, B1); memcpy(pi_a + B1, &y1, B1);               // :130-132
    c2.to_affine(pb, x2, y2); memcpy(pi_b, &x2, B2); memcpy(pi_b + B2, &y2, B2);
    c1.to_affine(pc, x1, y1); memcpy(pi_c, &x1, B1); memcpy(pi_c + B1, &y1, B1);
}

// The device part of one proof: the five MSM results of THIS key (shard) as Jacobian points jA | jB1 | jB2 | jC | jH
// (3*n8q bytes each, 6*n8q for jB2). With the full key they are the MSMs of src/groth16_prove.js:85-101; with a shard they are
// partial sums over its base-index range, to be added across devices before g16_finish.
// g16_enqueue puts the whole device part of a proof on the streams of the ACTIVE pipeline slot and returns without waiting;
// g16_complete waits for that slot and folds the window sums on the host.
// phase G16_ALL: the whole device part. The multi-GPU proof splits it so that no rank idles while the chain outputs travel:
// G16_W = the witness-side half (digit sorts of the witness, accumulations B2, B1, A, C, the G2 bucket reduction) — needs the witness only;
// G16_H = the H half (digit sort of this shard's H scalars, accumulation H, the batched G1 bucket reductions) — needs d_h_ext, computed
// elsewhere and received over xGMI. G16_W followed by G16_H in the same pipeline slot enqueues the same kernels as G16_ALL with d_h_ext.
enum { G16_ALL = 0, G16_W = 1, G16_H = 2 };
template <class FrC> static int g16_enqueue(G16Key& K, const void* d_witness, const void* d_h_ext = nullptr, int phase = G16_ALL) {
    Ctx& cx = ctx();
    ZK_TRY(g16_work_alloc(K, cx.pipe));
    G16Key::Work& Wk = K.wk[cx.pipe];
    if (Wk.in_flight) return fail(ZKMI_ERR_INVALID, "groth16: this pipeline slot already holds a proof in flight (collect it first)");
    if (phase == G16_H && !Wk.w_enqueued) return fail(ZKMI_ERR_INVALID, "groth16: the witness-side half of this proof has not been enqueued");
    if (phase != G16_H && Wk.w_enqueued) return fail(ZKMI_ERR_INVALID, "groth16: a witness-side half is waiting for its H half in this pipeline slot");
    if (phase == G16_H && !d_h_ext) return fail(ZKMI_ERR_INVALID, "groth16: the H half needs its scalars");
    const bool do_w = phase != G16_H, do_h = phase != G16_W, transforms = phase == G16_ALL && !d_h_ext;
    hipStream_t st = cx.stream;
    const uint32_t n = K.domain;
    const host::HField<4> Fr = host::HField<4>::from_cfg<FrC>();
    const uint32_t* w = (const uint32_t*)d_witness;
    const uint8_t* w_sh = (const uint8_t*)d_witness + (size_t)K.v_lo * 32;       // scalars of this shard's witness-side bases
    // d_h_ext: this shard's H-MSM scalars (h_cnt x 32 B, normal form) computed elsewhere (chain-parallel multi-GPU proof): the
    // buildABC / NTT / joinABC stages are skipped
    const uint8_t* h_sh = d_h_ext ? (const uint8_t*)d_h_ext : (const uint8_t*)Wk.T + (size_t)K.h_lo * 32;
    // The three digit sorts (witness without the B-infinity entries, witness, H scalars) are LDS/latency-bound; the NTT chain and
    // the bucket accumulations are ALU-bound. With ZKMI_OVERLAP (default) the sorts run on the auxiliary stream underneath them
    // and the main stream only waits on their events. ZKMI_OVERLAP=0 keeps everything on one stream.
    static const bool ov = !(getenv("ZKMI_OVERLAP") && atoi(getenv("ZKMI_OVERLAP")) == 0);
    MsmPlan pl, plh, plb;
    MsmJob* job = Wk.job;
    if (do_w) for (int i = 0; i < 5; i++) { job[i] = MsmJob(); ZK_TRY(msm_job_slot(i, job[i])); }
    // Two digit sorts of the witness: one without the entries whose B bases are at infinity (feeds B2 and B1), one complete
    // (feeds A and C). The second sort pays for itself once ~10 % of the B bases are at infinity.
    const bool split_b = K.b_density < 0.9;
    MsmJob* g2[1] = {&job[2]};
    hipStream_t aux = nullptr;
    if (ov) {
        ZK_TRY(msm_reduce_dispatch(K.curve, 2, g2, 0, true));        // njobs = 0: only creates the auxiliary stream
        aux = cx.aux_stream;
        if (!cx.sort_ev[0]) for (int i = 0; i < 5; i++) ZK_HIP(hipEventCreateWithFlags(&cx.sort_ev[i], hipEventDisableTiming));
    }
    if (ov && do_w) {
        ZK_HIP(hipEventRecord(cx.sort_ev[0], st));                   // the witness upload (if any) is ordered before this point
        ZK_HIP(hipStreamWaitEvent(aux, cx.sort_ev[0], 0));
        cx.stream = aux;
        int rc = msm_sort(w_sh, K.v_cnt, 32, split_b ? plb : pl, split_b ? 2 : 0, K.cw, 0, split_b ? K.drop_b : nullptr);
        if (!rc) rc = hipEventRecord(cx.sort_ev[1], aux) == hipSuccess ? 0 : fail(ZKMI_ERR_HIP, "hipEventRecord");
        if (!rc && split_b) rc = msm_sort(w_sh, K.v_cnt, 32, pl, 0, K.cw);
        if (!rc) rc = hipEventRecord(cx.sort_ev[2], aux) == hipSuccess ? 0 : fail(ZKMI_ERR_HIP, "hipEventRecord");
        cx.stream = st;
        ZK_TRY(rc);
    }
    if (do_w) ZK_HIP(hipEventRecord(Wk.ev[ST_BUILD], st));
    if (transforms) g16_build_abc<FrC>(K, Wk, w, st);
    if (do_w) ZK_HIP(hipEventRecord(Wk.ev[ST_NTT], st));
    if (transforms) {
        // inc = power == Fr.s ? Fr.shift : Fr.w[power+1] (:64); Fr.shift = nqr^2 — both come from the NTT module's root table
        uint8_t one[32], inc[32];
        memcpy(one, Fr.one, 32);
        ZK_TRY(fr_coset_inc(K.curve, K.power, inc));
        static const bool batch3 = !(getenv("ZKMI_NTT_BATCH") && atoi(getenv("ZKMI_NTT_BATCH")) == 0);
        if (batch3 && K.power > 0) {
            ZK_TRY(ntt_dev_batch_dispatch(K.curve, Wk.A, n, Wk.T, n, 3, K.power, 1, nullptr, nullptr));
            ZK_TRY(ntt_dev_batch_dispatch(K.curve, Wk.T, n, Wk.A, n, 3, K.power, 0, one, inc));
        } else {
            uint32_t* bufs[3] = {Wk.A, Wk.B, Wk.C};
            for (int k = 0; k < 3; k++) {
                ZK_TRY(ntt_dev_dispatch(K.curve, bufs[k], Wk.T, K.power, 1, nullptr, nullptr));
                ZK_TRY(ntt_dev_dispatch(K.curve, Wk.T, bufs[k], K.power, 0, one, inc));
            }
        }
    }
    if (do_w) ZK_HIP(hipEventRecord(Wk.ev[ST_JOIN], st));
    if (transforms) ZK_TRY(join_abc_dev_dispatch(K.curve, Wk.A, Wk.B, Wk.C, Wk.T, n));          // T = H-MSM scalars (normal form)
    if (do_w) ZK_HIP(hipEventRecord(Wk.ev[ST_SORT_W], st));
    // the digit sort of the H scalars: as soon as they exist (after joinABC here; at once when they come from outside)
    // In the split proof (G16_H after G16_W) the auxiliary stream still holds the G2 bucket reduction of the witness-side half (~2 ms of pure
    // latency): queued behind it the H sort — and with it the whole H half — would start that much later than in G16_ALL, where the sort was
    // enqueued BEFORE that reduction. The H sort of a split proof therefore gets a stream of its own (one per pipeline slot, highest priority).
    auto sort_h_aux = [&]() -> int {
        hipStream_t hs = aux;
        if (!do_w) {
            static hipStream_t h_sort_stream[2] = {nullptr, nullptr};
            if (!h_sort_stream[cx.pipe]) {
                int lo = 0, hi = 0;
                ZK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
                ZK_HIP(hipStreamCreateWithPriority(&h_sort_stream[cx.pipe], hipStreamNonBlocking, hi));
            }
            hs = h_sort_stream[cx.pipe];
        }
        ZK_HIP(hipEventRecord(cx.sort_ev[3], st));
        ZK_HIP(hipStreamWaitEvent(hs, cx.sort_ev[3], 0));
        cx.stream = hs;
        int rc = msm_sort(h_sh, K.h_cnt, 32, plh, 1, K.ch);
        if (!rc) rc = hipEventRecord(cx.sort_ev[4], hs) == hipSuccess ? 0 : fail(ZKMI_ERR_HIP, "hipEventRecord");
        cx.stream = st;
        return rc;
    };
    if (ov && do_h && do_w) ZK_TRY(sort_h_aux());
    if (do_w) {
    if (ov) ZK_HIP(hipStreamWaitEvent(st, cx.sort_ev[1], 0));
    else ZK_TRY(msm_sort(w_sh, K.v_cnt, 32, split_b ? plb : pl, split_b ? 2 : 0, K.cw, 0, split_b ? K.drop_b : nullptr));
    const MsmPlan& pB = split_b ? plb : pl;
    // The G2 MSM goes first: its bucket reduction is pure latency (~50 us per Fq2 point addition, little parallel work), so it
    // also runs on the auxiliary stream, underneath the G1 accumulations.
    ZK_HIP(hipEventRecord(Wk.ev[ST_MSM_B2], st));
    ZK_TRY(msm_accumulate_dispatch(K.curve, 2, K.bB2, pB, 0, job[2], K.mask[2]));
    if (ov) {
        ZK_HIP(hipEventRecord(cx.aux_ev[0], st));
        ZK_HIP(hipStreamWaitEvent(aux, cx.aux_ev[0], 0));
        ZK_TRY(msm_reduce_dispatch(K.curve, 2, g2, 1, true));
        ZK_HIP(hipEventRecord(cx.aux_ev[1], aux));
    }
    ZK_HIP(hipEventRecord(Wk.ev[ST_MSM_B1], st));
    ZK_TRY(msm_accumulate_dispatch(K.curve, 1, K.bB1, pB, 0, job[1], K.mask[1]));
    if (ov) ZK_HIP(hipStreamWaitEvent(st, cx.sort_ev[2], 0));
    else if (split_b) ZK_TRY(msm_sort(w_sh, K.v_cnt, 32, pl, 0, K.cw));
    ZK_HIP(hipEventRecord(Wk.ev[ST_MSM_A], st));
    ZK_TRY(msm_accumulate_dispatch(K.curve, 1, K.bA, pl, 0, job[0], K.mask[0]));
    ZK_HIP(hipEventRecord(Wk.ev[ST_MSM_C], st));
    ZK_TRY(msm_accumulate_dispatch(K.curve, 1, K.bC, pl, K.c_skip, job[3], K.mask[3]));
    ZK_HIP(hipEventRecord(Wk.ev[ST_SORT_H], st));
    Wk.pl_W = pl.sh.W; Wk.pl_nb = pl.sh.nb;
    }
    if (!do_h) {                                               // witness-side half only: the H half follows in this slot
        ZK_HIP(hipGetLastError());
        Wk.w_enqueued = true; Wk.ov = ov;
        return ZKMI_OK;
    }
    if (ov && !do_w) ZK_TRY(sort_h_aux());
    if (ov) ZK_HIP(hipStreamWaitEvent(st, cx.sort_ev[4], 0));
    else ZK_TRY(msm_sort(h_sh, K.h_cnt, 32, plh, 1, K.ch));
    ZK_HIP(hipEventRecord(Wk.ev[ST_MSM_H], st));
    // pi_c only needs C + H (:115): when both MSMs have the same bucket shape, H is accumulated into C's buckets and the two share
    // one bucket reduction (ZKMI_MERGE_CH=0 keeps them apart)
    static const bool merge_env = !(getenv("ZKMI_MERGE_CH") && atoi(getenv("ZKMI_MERGE_CH")) == 0);
    const bool merge_ch = merge_env && K.ch == K.cw && plh.sh.W == Wk.pl_W && plh.sh.nb == Wk.pl_nb;
    ZK_TRY(msm_accumulate_dispatch(K.curve, 1, K.bH, plh, 0, job[4], K.mask[4], merge_ch ? &job[3] : nullptr));
    ZK_HIP(hipEventRecord(Wk.ev[ST_REDUCE], st));
    // bucket reductions are latency-bound: all G1 jobs of one shape go through ONE set of launches
    MsmJob* g1[4] = {&job[0], &job[1], &job[3], &job[4]};
    if (merge_ch) ZK_TRY(msm_reduce_dispatch(K.curve, 1, g1, 3));
    else if (job[4].W == job[0].W && job[4].c == job[0].c) ZK_TRY(msm_reduce_dispatch(K.curve, 1, g1, 4));
    else { ZK_TRY(msm_reduce_dispatch(K.curve, 1, g1, 3)); ZK_TRY(msm_reduce_dispatch(K.curve, 1, g1 + 3, 1)); }
    if (!ov) ZK_TRY(msm_reduce_dispatch(K.curve, 2, g2, 1));
    ZK_HIP(hipEventRecord(Wk.ev[ST_COUNT], st));
    ZK_HIP(hipGetLastError());
    Wk.in_flight = true; Wk.ov = ov; Wk.w_enqueued = false;
    return ZKMI_OK;
}
template <class FrC> static int g16_complete(G16Key& K, uint8_t* jA, uint8_t* jB1, uint8_t* jB2, uint8_t* jC, uint8_t* jH) {
    Ctx& cx = ctx();
    G16Key::Work& Wk = K.wk[cx.pipe];
    if (!Wk.in_flight) return fail(ZKMI_ERR_INVALID, "groth16: no proof in flight in this pipeline slot");
    Wk.in_flight = false;
    MsmJob* job = Wk.job;
    ZK_HIP(hipStreamSynchronize(cx.stream));
    ZK_HIP(hipGetLastError());
    for (int i = 0; i < ST_COUNT; i++) { float ms = 0; if (hipEventElapsedTime(&ms, Wk.ev[i], Wk.ev[i + 1]) == hipSuccess) K.stage_ms[i] = ms; }
    // the host folds of the G1 jobs run while the G2 reduction may still be finishing on the auxiliary stream
    ZK_TRY(msm_fold_dispatch(K.curve, 1, job[0], jA)); ZK_TRY(msm_fold_dispatch(K.curve, 1, job[1], jB1));
    ZK_TRY(msm_fold_dispatch(K.curve, 1, job[3], jC)); ZK_TRY(msm_fold_dispatch(K.curve, 1, job[4], jH));
    if (Wk.ov) ZK_HIP(hipStreamSynchronize(cx.aux_stream));
    ZK_TRY(msm_fold_dispatch(K.curve, 2, job[2], jB2));
    return ZKMI_OK;
}
template <class FrC> static int g16_sums_dev(G16Key& K, const void* d_witness, uint8_t* jA, uint8_t* jB1, uint8_t* jB2, uint8_t* jC, uint8_t* jH) {
    ZK_TRY(g16_enqueue<FrC>(K, d_witness));
    return g16_complete<FrC>(K, jA, jB1, jB2, jC, jH);
}
static void g16_finish_dispatch(const G16Key& K, const uint8_t* jA, const uint8_t* jB1, const uint8_t* jB2, const uint8_t* jC, const uint8_t* jH, const uint8_t* r_mont,
                                const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c) {
    if (K.curve == ZKMI_CURVE_BN128) g16_finish<Fp<Bn254Fq>, Fp2<Bn254Fq>, Bn254Fr>(K, jA, jB1, jB2, jC, jH, r_mont, s_mont, pi_a, pi_b, pi_c);
    else g16_finish<Fp<Bls12381Fq>, Fp2<Bls12381Fq>, Bls12381Fr>(K, jA, jB1, jB2, jC, jH, r_mont, s_mont, pi_a, pi_b, pi_c);
}
template <class FrC> static int g16_prove_dev(G16Key& K, const void* d_witness, const uint8_t* r_mont, const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c) {
    if (!K.full()) return fail(ZKMI_ERR_INVALID, "groth16_prove: the key is a shard (use zkmi_groth16_sums_dev + zkmi_groth16_finish)");
    uint8_t jA[144], jB1[144], jB2[288], jC[144], jH[144];
    ZK_TRY(g16_sums_dev<FrC>(K, d_witness, jA, jB1, jB2, jC, jH));
    g16_finish_dispatch(K, jA, jB1, jB2, jC, jH, r_mont, s_mont, pi_a, pi_b, pi_c);
    return ZKMI_OK;
}

// buildABC + the selected iNTT -> coset -> NTT chains of one proof (bit 0: A, 1: B, 2: C), results copied out of the key's work
// buffers: the chain-parallel part of a multi-GPU proof (each chain on a different rank)
template <class FrC> static int g16_chains(G16Key& K, const void* d_witness, unsigned mask, void* d_a, void* d_b, void* d_c) {
    Ctx& cx = ctx();
    hipStream_t st = cx.stream;
    ZK_TRY(g16_work_alloc(K, cx.pipe));
    G16Key::Work& Wk = K.wk[cx.pipe];
    if (Wk.in_flight) return fail(ZKMI_ERR_INVALID, "groth16_chains: a proof is in flight in this pipeline slot");
    const host::HField<4> Fr = host::HField<4>::from_cfg<FrC>();
    const uint32_t n = K.domain;
    g16_build_abc<FrC>(K, Wk, (const uint32_t*)d_witness, st);
    uint8_t one[32], inc[32];
    memcpy(one, Fr.one, 32);
    ZK_TRY(fr_coset_inc(K.curve, K.power, inc));
    uint32_t* bufs[3] = {Wk.A, Wk.B, Wk.C};
    void* outs[3] = {d_a, d_b, d_c};
    for (int k = 0; k < 3; k++) {
        if (!((mask >> k) & 1u)) continue;
        if (!outs[k]) return fail(ZKMI_ERR_INVALID, "groth16_chains: null output for a selected chain");
        ZK_TRY(ntt_dev_dispatch(K.curve, bufs[k], Wk.T, K.power, 1, nullptr, nullptr));
        ZK_TRY(ntt_dev_dispatch(K.curve, Wk.T, outs[k], K.power, 0, one, inc));
    }
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
static G16Key* g16_find(uint64_t key) {
    auto it = ctx().groth16.find(key);
    return it == ctx().groth16.end() ? nullptr : (G16Key*)it->second;
}
static uint64_t g_last_key = 0;

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zkmi_groth16_load(const zkmi_groth16_zkey* zkey, uint64_t key) {
    ZK_TRY(require_ctx());
    if (!zkey || !key) return fail(ZKMI_ERR_INVALID, "groth16_load: null zkey or key 0");
    return g16_load(zkey, key, 0, zkey->n_vars, 0, zkey->domain_size);
}
int zkmi_groth16_load_shard(const zkmi_groth16_zkey* zkey, uint64_t key, uint32_t var_lo, uint32_t var_hi, uint32_t h_lo, uint32_t h_hi) {
    ZK_TRY(require_ctx());
    if (!zkey || !key) return fail(ZKMI_ERR_INVALID, "groth16_load_shard: null zkey or key 0");
    return g16_load(zkey, key, var_lo, var_hi, h_lo, h_hi);
}
int zkmi_groth16_load_paged(const zkmi_groth16_zkey_paged* zkey, uint64_t key) {
    ZK_TRY(require_ctx());
    if (!zkey || !key) return fail(ZKMI_ERR_INVALID, "groth16_load: null zkey or key 0");
    return g16_load(zkey, key, 0, zkey->n_vars, 0, zkey->domain_size);
}
int zkmi_groth16_load_shard_paged(const zkmi_groth16_zkey_paged* zkey, uint64_t key, uint32_t var_lo, uint32_t var_hi, uint32_t h_lo, uint32_t h_hi) {
    ZK_TRY(require_ctx());
    if (!zkey || !key) return fail(ZKMI_ERR_INVALID, "groth16_load_shard: null zkey or key 0");
    return g16_load(zkey, key, var_lo, var_hi, h_lo, h_hi);
}
int zkmi_groth16_build_abc_dev(uint64_t key, const void* d_witness, void* d_a, void* d_b, void* d_c) {
    ZK_TRY(require_ctx());
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_build_abc_dev: key not loaded");
    if (!d_witness) return fail(ZKMI_ERR_INVALID, "groth16_build_abc_dev: null witness");
    Ctx& cx = ctx();
    ZK_TRY(g16_work_alloc(*K, cx.pipe));
    G16Key::Work& Wk = K->wk[cx.pipe];
    if (Wk.in_flight || Wk.w_enqueued) return fail(ZKMI_ERR_INVALID, "groth16_build_abc_dev: a proof is in flight in this pipeline slot");
    ZK_HIP(hipEventRecord(cx.ev0, cx.stream));
    if (K->curve == ZKMI_CURVE_BN128) g16_build_abc<Bn254Fr>(*K, Wk, (const uint32_t*)d_witness, cx.stream);
    else g16_build_abc<Bls12381Fr>(*K, Wk, (const uint32_t*)d_witness, cx.stream);
    ZK_HIP(hipEventRecord(cx.ev1, cx.stream));
    const size_t bytes = (size_t)K->domain * 32;
    if (d_a) ZK_HIP(hipMemcpyAsync(d_a, Wk.A, bytes, hipMemcpyDeviceToDevice, cx.stream));
    if (d_b) ZK_HIP(hipMemcpyAsync(d_b, Wk.B, bytes, hipMemcpyDeviceToDevice, cx.stream));
    if (d_c) ZK_HIP(hipMemcpyAsync(d_c, Wk.C, bytes, hipMemcpyDeviceToDevice, cx.stream));
    ZK_HIP(hipStreamSynchronize(cx.stream));
    ZK_HIP(hipGetLastError());
    float ms = 0;
    if (hipEventElapsedTime(&ms, cx.ev0, cx.ev1) == hipSuccess) cx.last_ms = ms;
    return ZKMI_OK;
}
int zkmi_groth16_coef_layout(uint64_t key, uint64_t* out, int n) {
    const G16Key* K = g16_find(key);
    if (!K || !out) return fail(ZKMI_ERR_INVALID, "groth16_coef_layout: key not loaded");
    const uint64_t v[5] = {K->n_coef, K->n_seg, K->n_long, K->n_part, (uint64_t)K->sell_terms};
    for (int i = 0; i < n && i < 5; i++) out[i] = v[i];
    return ZKMI_OK;
}
int zkmi_groth16_sums_dev(uint64_t key, const void* d_witness, uint8_t* sums) {
    ZK_TRY(require_ctx());
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_sums_dev: key not loaded");
    if (!d_witness || !sums) return fail(ZKMI_ERR_INVALID, "groth16_sums_dev: null argument");
    g_last_key = key;
    const size_t j1 = 3 * (size_t)n8q_of(K->curve);
    uint8_t *jA = sums, *jB1 = sums + j1, *jB2 = sums + 2 * j1, *jC = sums + 4 * j1, *jH = sums + 5 * j1;
    if (K->curve == ZKMI_CURVE_BN128) return g16_sums_dev<Bn254Fr>(*K, d_witness, jA, jB1, jB2, jC, jH);
    return g16_sums_dev<Bls12381Fr>(*K, d_witness, jA, jB1, jB2, jC, jH);
}
int zkmi_groth16_finish(uint64_t key, const uint8_t* sums, const uint8_t* r_mont, const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c) {
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_finish: key not loaded");
    if (!sums || !r_mont || !s_mont || !pi_a || !pi_b || !pi_c) return fail(ZKMI_ERR_INVALID, "groth16_finish: null argument");
    const size_t j1 = 3 * (size_t)n8q_of(K->curve);
    g16_finish_dispatch(*K, sums, sums + j1, sums + 2 * j1, sums + 4 * j1, sums + 5 * j1, r_mont, s_mont, pi_a, pi_b, pi_c);
    return ZKMI_OK;
}
int zkmi_groth16_release(uint64_t key) {
    auto it = ctx().groth16.find(key);
    if (it == ctx().groth16.end()) return ZKMI_OK;
    if (ctx().ready) (void)hipDeviceSynchronize();            // both pipeline slots, main and auxiliary streams
    delete (G16Key*)it->second;
    ctx().groth16.erase(it);
    return ZKMI_OK;
}
int zkmi_groth16_prove_dev(uint64_t key, const void* d_witness, const uint8_t* r_mont, const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c) {
    ZK_TRY(require_ctx());
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_prove_dev: key not loaded");
    if (!d_witness || !r_mont || !s_mont || !pi_a || !pi_b || !pi_c) return fail(ZKMI_ERR_INVALID, "groth16_prove_dev: null argument");
    g_last_key = key;
    if (K->curve == ZKMI_CURVE_BN128) return g16_prove_dev<Bn254Fr>(*K, d_witness, r_mont, s_mont, pi_a, pi_b, pi_c);
    return g16_prove_dev<Bls12381Fr>(*K, d_witness, r_mont, s_mont, pi_a, pi_b, pi_c);
}
int zkmi_groth16_chains_dev(uint64_t key, const void* d_witness, unsigned chain_mask, void* d_a, void* d_b, void* d_c) {
    ZK_TRY(require_ctx());
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_chains_dev: key not loaded");
    if (!d_witness || chain_mask > 7u) return fail(ZKMI_ERR_INVALID, "groth16_chains_dev: bad argument");
    if (K->curve == ZKMI_CURVE_BN128) return g16_chains<Bn254Fr>(*K, d_witness, chain_mask, d_a, d_b, d_c);
    return g16_chains<Bls12381Fr>(*K, d_witness, chain_mask, d_a, d_b, d_c);
}
int zkmi_groth16_sums_w_dev(uint64_t key, const void* d_witness) {
    ZK_TRY(require_ctx());
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_sums_w_dev: key not loaded");
    if (!d_witness) return fail(ZKMI_ERR_INVALID, "groth16_sums_w_dev: null witness");
    g_last_key = key;
    return K->curve == ZKMI_CURVE_BN128 ? g16_enqueue<Bn254Fr>(*K, d_witness, nullptr, G16_W) : g16_enqueue<Bls12381Fr>(*K, d_witness, nullptr, G16_W);
}
int zkmi_groth16_sums_h_dev(uint64_t key, const void* d_witness, const void* d_h_scalars, uint8_t* sums) {
    ZK_TRY(require_ctx());
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_sums_h_dev: key not loaded");
    if (!d_witness || !d_h_scalars || !sums) return fail(ZKMI_ERR_INVALID, "groth16_sums_h_dev: null argument");
    g_last_key = key;
    const size_t j1 = 3 * (size_t)n8q_of(K->curve);
    uint8_t *jA = sums, *jB1 = sums + j1, *jB2 = sums + 2 * j1, *jC = sums + 4 * j1, *jH = sums + 5 * j1;
    // after zkmi_groth16_sums_w_dev only the H half is left to enqueue; without it the call runs both halves
    const int phase = K->wk[ctx().pipe].w_enqueued ? G16_H : G16_ALL;
    if (K->curve == ZKMI_CURVE_BN128) { ZK_TRY(g16_enqueue<Bn254Fr>(*K, d_witness, d_h_scalars, phase)); return g16_complete<Bn254Fr>(*K, jA, jB1, jB2, jC, jH); }
    ZK_TRY(g16_enqueue<Bls12381Fr>(*K, d_witness, d_h_scalars, phase));
    return g16_complete<Bls12381Fr>(*K, jA, jB1, jB2, jC, jH);
}
/* Host-witness variant of zkmi_groth16_submit_dev: the witness crosses PCIe on the SLOT's stream into the slot's own buffer, so the upload of
 * proof k+1 runs underneath the kernels of proof k in the other slot (throughput mode of a host that holds witnesses in host memory: the
 * Node addon). `witness` must stay valid until the call returns (pageable memory is staged by the runtime before it does). */
int zkmi_groth16_submit(uint64_t key, const uint8_t* witness, size_t witness_len, int slot) {
    ZK_TRY(require_ctx());
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_submit: key not loaded");
    if (!K->full()) return fail(ZKMI_ERR_INVALID, "groth16_submit: the key is a shard");
    if (!witness) return fail(ZKMI_ERR_INVALID, "groth16_submit: null witness");
    if (witness_len != (size_t)K->n_vars * 32)
        return fail(ZKMI_ERR_INVALID, "Invalid witness length. Circuit: " + std::to_string(K->n_vars) + ", witness: " + std::to_string(witness_len / 32));
    ZK_TRY(select_pipe(slot));
    g_last_key = key;
    int rc = g16_work_alloc(*K, slot);
    if (!rc && K->wk[slot].in_flight) rc = fail(ZKMI_ERR_INVALID, "groth16: this pipeline slot already holds a proof in flight (collect it first)");
    if (!rc && hipMemcpyAsync(K->wk[slot].w, witness, witness_len, hipMemcpyHostToDevice, ctx().stream) != hipSuccess) rc = fail(ZKMI_ERR_HIP, "groth16_submit: witness upload");
    if (!rc) rc = K->curve == ZKMI_CURVE_BN128 ? g16_enqueue<Bn254Fr>(*K, K->wk[slot].w) : g16_enqueue<Bls12381Fr>(*K, K->wk[slot].w);
    int rc2 = select_pipe(0);
    return rc ? rc : rc2;
}
int zkmi_groth16_submit_dev(uint64_t key, const void* d_witness, int slot) {
    ZK_TRY(require_ctx());
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_submit_dev: key not loaded");
    if (!K->full()) return fail(ZKMI_ERR_INVALID, "groth16_submit_dev: the key is a shard");
    if (!d_witness) return fail(ZKMI_ERR_INVALID, "groth16_submit_dev: null witness");
    ZK_TRY(select_pipe(slot));
    g_last_key = key;
    int rc = K->curve == ZKMI_CURVE_BN128 ? g16_enqueue<Bn254Fr>(*K, d_witness) : g16_enqueue<Bls12381Fr>(*K, d_witness);
    int rc2 = select_pipe(0);
    return rc ? rc : rc2;
}
int zkmi_groth16_collect(uint64_t key, int slot, const uint8_t* r_mont, const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c) {
    ZK_TRY(require_ctx());
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_collect: key not loaded");
    if (!r_mont || !s_mont || !pi_a || !pi_b || !pi_c) return fail(ZKMI_ERR_INVALID, "groth16_collect: null argument");
    ZK_TRY(select_pipe(slot));
    uint8_t jA[144], jB1[144], jB2[288], jC[144], jH[144];
    int rc = K->curve == ZKMI_CURVE_BN128 ? g16_complete<Bn254Fr>(*K, jA, jB1, jB2, jC, jH) : g16_complete<Bls12381Fr>(*K, jA, jB1, jB2, jC, jH);
    int rc2 = select_pipe(0);
    if (rc || rc2) return rc ? rc : rc2;
    g16_finish_dispatch(*K, jA, jB1, jB2, jC, jH, r_mont, s_mont, pi_a, pi_b, pi_c);
    return ZKMI_OK;
}
static int g16_prove_host(const zkmi_groth16_zkey_paged* zkey, uint64_t key, const uint8_t* witness, size_t witness_len, const uint8_t* r_mont, const uint8_t* s_mont,
                          uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c) {
    ZK_TRY(require_ctx());
    if (!witness) return fail(ZKMI_ERR_INVALID, "groth16_prove: null witness");
    const uint64_t k = key ? key : 0xffffffffffffffffull;          // key 0: load, prove, release
    if (!g16_find(k) || !key) {
        if (!zkey) return fail(ZKMI_ERR_INVALID, "groth16_prove: key not loaded and no zkey given");
        ZK_TRY(g16_load(zkey, k, 0, zkey->n_vars, 0, zkey->domain_size));
    } else if (zkey) {
        // a descriptor next to a resident key must describe the same circuit: a caller that re-uses key numbers for different
        // zkeys would otherwise get a proof for the FIRST circuit with rc 0
        const G16Key* R = g16_find(k);
        const size_t coeffs_len = pg_total(zkey->coeffs);
        if (R->curve != zkey->curve || R->n_vars != zkey->n_vars || R->n_public != zkey->n_public || R->domain != zkey->domain_size ||
            coeffs_len < 4 || (size_t)R->n_coef != (coeffs_len - 4) / 44)
            return fail(ZKMI_ERR_INVALID, "groth16_prove: the resident key under this cache key belongs to a different circuit (release it first)");
        // same shape is not the same key: another phase-2 contribution of the same circuit has other delta / bases. The header points are cheap
        // to compare and change with every contribution
        const size_t g1b = 2 * (size_t)n8q_of(R->curve), g2b = 2 * g1b;
        if (!zkey->vk_alpha_1 || !zkey->vk_beta_1 || !zkey->vk_beta_2 || !zkey->vk_delta_1 || !zkey->vk_delta_2 || memcmp(R->vk_alpha_1.data(), zkey->vk_alpha_1, g1b) ||
            memcmp(R->vk_beta_1.data(), zkey->vk_beta_1, g1b) || memcmp(R->vk_beta_2.data(), zkey->vk_beta_2, g2b) || memcmp(R->vk_delta_1.data(), zkey->vk_delta_1, g1b) ||
            memcmp(R->vk_delta_2.data(), zkey->vk_delta_2, g2b))
            return fail(ZKMI_ERR_INVALID, "groth16_prove: the resident key under this cache key is another key of the same circuit shape (different alpha / beta / delta: release it first)");
    }
    G16Key* K = g16_find(k);
    if (witness_len != (size_t)K->n_vars * 32) {
        const std::string msg = "Invalid witness length. Circuit: " + std::to_string(K->n_vars) + ", witness: " + std::to_string(witness_len / 32);
        if (!key) zkmi_groth16_release(k);
        return fail(ZKMI_ERR_INVALID, msg);
    }
    ZK_TRY(select_pipe(0));
    ZK_HIP(hipMemcpyAsync(K->wk[0].w, witness, (size_t)K->n_vars * 32, hipMemcpyHostToDevice, ctx().stream));
    int rc = zkmi_groth16_prove_dev(k, K->wk[0].w, r_mont, s_mont, pi_a, pi_b, pi_c);
    if (!key) zkmi_groth16_release(k);
    return rc;
}
int zkmi_groth16_prove(const zkmi_groth16_zkey* zkey, uint64_t key, const uint8_t* witness, size_t witness_len, const uint8_t* r_mont, const uint8_t* s_mont,
                       uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c) {
    if (!zkey) return g16_prove_host(nullptr, key, witness, witness_len, r_mont, s_mont, pi_a, pi_b, pi_c);
    FlatAsPaged f(*zkey);
    return g16_prove_host(&f.pz, key, witness, witness_len, r_mont, s_mont, pi_a, pi_b, pi_c);
}
int zkmi_groth16_prove_paged(const zkmi_groth16_zkey_paged* zkey, uint64_t key, const uint8_t* witness, size_t witness_len, const uint8_t* r_mont, const uint8_t* s_mont,
                             uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c) {
    return g16_prove_host(zkey, key, witness, witness_len, r_mont, s_mont, pi_a, pi_b, pi_c);
}
int zkmi_groth16_key_curve(uint64_t key) {
    const G16Key* K = g16_find(key);
    return K ? K->curve : -1;
}
int zkmi_groth16_reset(uint64_t key) {
    G16Key* K = g16_find(key);
    if (!K) return fail(ZKMI_ERR_INVALID, "groth16_reset: key not loaded");
    if (ctx().ready) ZK_HIP(hipDeviceSynchronize());            // both pipeline slots, main and auxiliary streams
    for (auto& w : K->wk) { w.in_flight = false; w.w_enqueued = false; }
    return ZKMI_OK;
}
int zkmi_groth16_stage_ms(double* out, int n) {
    G16Key* K = g16_find(g_last_key);
    if (!K || !out) return fail(ZKMI_ERR_INVALID, "groth16_stage_ms: no proof has been run");
    for (int i = 0; i < n && i < ST_COUNT; i++) out[i] = K->stage_ms[i];
    return ZKMI_OK;
}

}  // extern "C"
